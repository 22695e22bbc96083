"""Neighbor lists: the data ``hoomd.md.nlist.Cell(buffer=...)`` hands to the pair
potentials in every reference test (e.g. src/pytest/test_pair.py:337), built on
the GPU by libazp's ``azp_nlist_*`` kernels.

torch is used for the sort / scan steps between the kernels and to hold the
buffers; the binning, counting and filling are libazp kernels.
"""

import ctypes as C
import os

import numpy as np

from . import _lib


class NeighborList:
    """Base: holds the device arrays in HOOMD's layout."""

    def __init__(self, buffer, exclusions=("bond",)):
        self.buffer = float(buffer)
        self.exclusions = tuple(exclusions)
        self._consumers = []
        self.n_neigh = None
        self._head_list = None
        self._nlist = None
        self._size = 0
        self._built_generation = None
        self.num_builds = 0
        self._n_pairs = 0         # listed pairs = sum(n_neigh)
        self._max_neigh = 0
        self._row_capacity = 0    # > 0: rows of fixed capacity (single-pass rebuilds)
        self.single_pass = True
        # fused = True: when the list has ONE consumer and that consumer runs the tile kernels, no
        # HOOMD-format list is built at all -- the consumer compiles its tile plan straight from
        # the binned particles (azp_pair_plan_build_from_cells); nlist / head_list are then filled
        # in lazily if something asks for them
        self.fused = True
        self._fused_active = False
        self._stats_known = True
        self._consumer_version = 0        # bumped when a consumer's r_cut matrix changes
        self._built_consumer_version = None
        # domain-decomposed runs (azplugins_amd.domain): the rebuild decision is collective
        # (reduce_flag: bool -> bool, an all-reduce over the ranks) and particles migrate /
        # ghosts are re-selected before the list is rebuilt (before_rebuild(state))
        self.reduce_flag = None
        self.before_rebuild = None

    # -- consumers (pair potentials) register their r_cut matrices ---------
    def _add_consumer(self, force):
        if force not in self._consumers:
            self._consumers.append(force)
            self._consumers_changed()

    def _consumers_changed(self):
        """A consumer was added or changed its r_cut: the next compute rebuilds the list."""
        self._consumer_version += 1

    def _r_cut_matrix(self, ntypes):
        rc = np.zeros((ntypes, ntypes))
        for f in self._consumers:
            rc = np.maximum(rc, f._r_cut_matrix())
        return rc

    # -- the HOOMD-format arrays; in fused mode they are built on first use ------------
    @property
    def nlist(self):
        self._materialize()
        return self._nlist

    @property
    def head_list(self):
        self._materialize()
        return self._head_list

    @property
    def size(self):
        self._materialize()
        return self._size

    def _materialize(self):
        pass

    def _stats(self):
        """``n_pairs`` / ``max_neigh``. In fused mode (the plan compiled straight from the cells) the row lengths come
        from the plan compiler, whose single-precision acceptance test keeps a superset of the exact list by a hair
        (a few 1e-5 of the entries, all of them buffer entries the force kernel's exact cutoff test ignores): the
        statistics are then UPPER BOUNDS of HOOMD's exact counts, and become exact once the list is materialized."""
        if not self._stats_known and self.n_neigh is not None:
            import torch

            if self._fused_active and not getattr(self, "_fused_counts_ready", False):
                self._materialize()  # nobody has compiled a plan from the cells yet: count the classic way
                return

            n = self.n_neigh.to(torch.int64)
            self._max_neigh, self._n_pairs = (int(v) for v in torch.stack([n.max(), n.sum()]).tolist()) if n.numel() else (0, 0)
            self._stats_known = True

    @property
    def n_pairs(self):
        self._stats()
        return self._n_pairs

    @property
    def max_neigh(self):
        self._stats()
        return self._max_neigh

    @property
    def built(self):
        return self.n_neigh is not None

    @property
    def r_list_max(self):
        """Upper bound on the separation of any listed pair between rebuilds."""
        return self._r_cut_max + 2.0 * self.buffer


class Cell(NeighborList):
    """Cell-list neighbor list (full storage, as HOOMD's GPU pair kernels use)."""

    # azp_nlist_bin (counting sort in libazp) instead of the framework's sort pipeline: by default (None) where the cells
    # number more than 2^16 (16-bit keys do not hold them and the framework sort needs four radix passes), AZP_NATIVE_BINNING=1 / 0
    # forces it on / off. For the north star's 2^15 cells the two are equal in a 300-step MD run (0.402-0.403 against
    # 0.403-0.406 ms per step at the end of round 3; an earlier version of the library's kernels measured slower); kept also for
    # callers without the framework, tested for equality with the stable sort.
    native_binning = {"0": False, "1": True}.get(os.environ.get("AZP_NATIVE_BINNING", ""), None)  # None: when the cells number more than 2^16 (the framework sort then needs four radix passes)

    def compute(self, state, force=False, compact=False):
        """Rebuild only when needed (HOOMD's criterion): never built, forced, or some
        particle moved farther than buffer / 2 since the last build.

        The first build counts, scans and fills (exact rows). Later builds reuse the
        row capacity learned from the previous one and run the fill alone (HOOMD's
        protocol: fixed-capacity rows, rebuilt when a row overflows);
        ``compact=True`` forces exact rows."""
        self._compact = compact
        if getattr(state, "order_generation", 0) != getattr(self, "_order_generation", 0):
            force = True  # the particles were re-indexed (ParticleSorter): every stored index is stale
        if self._built_consumer_version != self._consumer_version:
            force = True  # built for another r_cut matrix
        if not force and self.built:
            if self._built_generation == state.position_generation:
                return
            v = getattr(self, "_verdict", None)
            # (a caller that queued its kernel behind the check -- Pair._compute_speculative -- has the verdict already)
            moved = v[1] if (v is not None and v[0] == state.position_generation) else self._moved_too_far(state)
            if self.reduce_flag is not None:
                moved = bool(self.reduce_flag(moved))
            if not moved:
                self._built_generation = state.position_generation
                return
            if self.before_rebuild is not None:
                self.before_rebuild(state)
        self._build(state)
        self._built_generation = state.position_generation

    def _moved_too_far(self, state):
        """One kernel + an 16-byte readback (HOOMD's distance check); also records the
        largest displacement since the build (``displacement_bound``)."""
        return self.end_check(self.begin_check(state))

    def begin_check(self, state):
        """Queue the distance check on the current stream and its 16-byte readback on a side stream (ordered after
        the check alone). Returns a token for ``end_check``; ``token["flag_ptr"]`` / ``token["bits_ptr"]`` are the
        device words a force kernel queued right behind the check can read (azp_pair_args.d_stale_flag,
        d_displacement_sq_bits), so that the host's wait for the result does not idle the GPU."""
        import torch

        # [flag, max |dx|^2 bits] per check, a ring of 64 rows zeroed once per 64 checks (not a fill kernel per step)
        if getattr(self, "_flag", None) is None or self._flag.device != state.pos.device:
            self._flag = torch.zeros((64, 2), dtype=torch.int64, device=state.pos.device)
            self._flag_i = 0
            self._side = torch.cuda.Stream(device=state.pos.device)
            self._host_row = torch.zeros(2, dtype=torch.int64).pin_memory()
        if self._flag_i == 64:
            self._flag.zero_()
            self._flag_i = 0
        row = self._flag[self._flag_i]
        self._flag_i += 1
        box = state.box.to_c()
        stream = _lib.raw_stream(state.device)
        # every particle's own displacement next to the maximum: the tile kernels can take the maximum over what a
        # tile stages (azp_pair_args.d_displacement) instead of the global one
        if getattr(self, "_disp_arr", None) is None or self._disp_arr.shape[0] != state.n_max or self._disp_arr.device != state.pos.device:
            self._disp_arr = torch.zeros(state.n_max, dtype=torch.float32, device=state.pos.device)
        _lib.check(_lib.lib().azp_nlist_displacements(state.n_max, state.pos.data_ptr(), self._pos_at_build.data_ptr(),
                                                      C.byref(box), (0.5 * self.buffer) ** 2, row.data_ptr(),
                                                      row.data_ptr() + 8, self._disp_arr.data_ptr(), stream),
                   "azp_nlist_displacements")
        ring = self.__dict__.get("_done_events")
        if ring is None:
            ring = self._done_events = [torch.cuda.Event() for _ in range(4)]  # (recorded and waited for within one step)
        done = ring[self._flag_i & 3]
        done.record()
        with torch.cuda.stream(self._side):
            self._side.wait_event(done)
            self._host_row.copy_(row, non_blocking=True)
        return dict(flag_ptr=row.data_ptr(), bits_ptr=row.data_ptr() + 8, generation=state.position_generation, row=row)

    def end_check(self, token):
        """Wait for the check of ``begin_check`` (not for anything queued after it); True: rebuild."""
        self._side.synchronize()
        flag, bits = self._host_row.tolist()
        self._disp = float(np.sqrt(np.array([bits], dtype=np.int64).view(np.float64)[0]))
        self._disp_generation = token["generation"]
        self._disp_arr_generation = token["generation"]
        self._verdict = (token["generation"], bool(flag))
        return bool(flag)

    def displacement_bound(self, state):
        """Largest distance any particle has moved since the list was built, if known
        for the current positions (else None)."""
        if getattr(self, "_disp_generation", None) == state.position_generation:
            return self._disp
        return None

    def displacements(self, state):
        """Per-particle displacements since the list was built (float32 device tensor of n_max upper
        bounds), if known for the current positions (else None)."""
        if getattr(self, "_disp_arr_generation", None) == state.position_generation and getattr(self, "_disp_arr", None) is not None:
            return self._disp_arr
        return None

    def assume_displacement(self, state, bound, per_particle=None):
        """Benchmark / replay hook: the caller vouches that the current positions are within
        ``bound`` of the positions the list was built for (e.g. a stored snapshot of a run whose
        distance check returned exactly that), so no distance check runs for them.
        ``per_particle``: that check's per-particle displacements (a clone of ``displacements()``)."""
        if not self.built:
            raise _lib.AzpError("assume_displacement: the list has not been built")
        self._built_generation = state.position_generation
        self._disp, self._disp_generation = float(bound), state.position_generation
        if per_particle is not None:
            self._disp_arr, self._disp_arr_generation = per_particle, state.position_generation
        else:
            self._disp_arr_generation = None

    def _build(self, state):
        import torch

        l = _lib.lib()
        dev = state.device
        ntypes = len(state.types)
        rc = self._r_cut_matrix(ntypes)
        self._r_cut_max = float(rc.max())
        rl = np.where(rc > 0.0, rc + self.buffer, 0.0)
        rl_max = float(rl.max())
        if rl_max <= 0.0:
            raise _lib.AzpError("neighbor list has no consumer with r_cut > 0")
        box = state.box
        L = box.L
        for k in range(3):
            if box.periodic[k] and L[k] < 2.0 * rl_max:
                raise _lib.AzpError("box dimension %d (%g) is smaller than 2 (r_cut + buffer) = %g" % (k, L[k], 2 * rl_max))
        n_total = state.n_max
        N = state.N
        a = _lib.NlistArgs()
        a.N = N
        a.n_total = n_total
        a.d_pos = state.pos.data_ptr()
        a.box = box.to_c()
        # grid over the periodic box; a triclinic box is binned in its bounding
        # orthorhombic frame only when untilted, so require orthorhombic here
        if box.is_triclinic:
            raise _lib.AzpError("Cell neighbor list: triclinic boxes are not supported yet")
        a.ntypes = ntypes
        # (kept across rebuilds while the cutoffs stand: a host-to-device copy from pageable memory waits for the stream)
        key = (self._consumer_version, self.buffer, str(dev), ntypes)
        if getattr(self, "_rlistsq_key", None) != key:
            self._rlistsq_dev = torch.from_numpy(np.ascontiguousarray((rl * rl).reshape(-1))).to(dev)
            self._rlistsq_key = key
        rlistsq = self._rlistsq_dev
        a.d_rlistsq = rlistsq.data_ptr()
        stream = torch.cuda.current_stream(dev).cuda_stream
        self._rl_max, self._box_at_build = rl_max, box
        self._fused_active = self._fused_eligible()
        sub = 2 if (self._fused_active and self._half_cells_wanted(box, rl_max, n_total)) else 1
        bins = self._bin(a, box, rl_max, n_total, sub, dev, stream)

        keep = []
        if "bond" in self.exclusions and state.n_bonds:
            n_excl, excl, pitch = state.exclusion_table()
            a.d_n_excl = n_excl.data_ptr()
            a.d_excl = excl.data_ptr()
            a.excl_pitch = pitch
            keep += [n_excl, excl]

        n_neigh = torch.empty(N, dtype=torch.int32, device=dev)
        a.d_n_neigh = n_neigh.data_ptr()
        self.n_neigh = n_neigh
        self._cells = a
        self._keep = (rlistsq, bins, keep)
        self._pos_at_build = state.pos[:n_total].clone()
        self._nlist, self._head_list, self._size = None, None, 0
        self._fused_counts_ready = False
        if self._fused_active:
            self._stats_known = False  # row lengths come from the consumer's plan compile
        else:
            self._fill(stream)
        self._order_generation = getattr(state, "order_generation", 0)
        self._built_consumer_version = self._consumer_version
        self._disp, self._disp_generation = 0.0, state.position_generation
        self.num_builds += 1

    # Cells of half the list radius for the fused plan compile (azp_nlist_args.cell_subdivision = 2, csrc/pair_plan_cells.hip):
    # 0 never (default), 1 when the geometry suits them (a box of >= 8 such cells along every periodic axis, >= 0.75
    # particles per cell), 2 whenever the compiler can take them (>= 5 cells along every periodic axis; tests).
    # Measured on the north-star liquid: 338 instead of 864 candidate tests per particle, but the same
    # 1.7 ms per build -- with every lane on its own candidates the waves lose the lock-step of the full-width form (all
    # members of a cell walk the same candidates, and a pair of candidates nobody in the wave accepts skips the accept
    # path), and the finer binning costs more. Kept as an exact, tested alternative; off by default.
    half_cells = int(os.environ.get("AZP_HALF_CELLS", "0"))

    def _half_cells_wanted(self, box, rl_max, n_total):
        mode = self.half_cells
        if mode <= 0 or getattr(self, "_half_failures", 0) >= 2:
            return False
        L = box.L
        dims = [max(int(np.floor(L[k] / (0.5 * rl_max))), 1) for k in range(3)]
        need = 5 if mode >= 2 else 8
        if any(box.periodic[k] and dims[k] < need for k in range(3)):
            return False
        return mode >= 2 or n_total >= 0.75 * dims[0] * dims[1] * dims[2]

    def _bin(self, a, box, rl_max, n_total, sub, dev, stream):
        """Bin the particles of ``a.d_pos`` into cells at least ``rl_max / sub`` wide: fills the grid, d_cell_of,
        d_order, d_cell_start of ``a``; returns the tensors to keep alive."""
        import torch

        l = _lib.lib()
        L = box.L
        for k in range(3):
            dim = max(int(np.floor(L[k] / (rl_max / sub))), 1)
            a.grid.dim[k] = dim
            a.grid.width[k] = L[k] / dim
            a.grid.lo[k] = -0.5 * L[k]
            a.grid.periodic[k] = 1 if box.periodic[k] else 0
        a.cell_subdivision = sub
        ncell = int(a.grid.dim[0]) * int(a.grid.dim[1]) * int(a.grid.dim[2])
        cell_of = torch.empty(n_total, dtype=torch.int32, device=dev)
        a.d_cell_of = cell_of.data_ptr()
        native = self.native_binning if self.native_binning is not None else (ncell > 65536)
        if native and n_total <= 128 * ncell:
            # one libazp call (counting sort, stable in the particle index): five small kernels back to back instead
            # of a framework sort pipeline of a dozen launches with host gaps between them. Cells that hold very many
            # particles (tiny boxes) take the general path: the per-cell sort is quadratic in the cell's population.
            order = torch.empty(n_total, dtype=torch.int32, device=dev)
            cell_start = torch.empty(ncell + 1, dtype=torch.int32, device=dev)
            cursor = torch.empty(ncell, dtype=torch.int32, device=dev)
            order_tmp = torch.empty(n_total, dtype=torch.int32, device=dev)
            a.d_order = order.data_ptr()
            a.d_cell_start = cell_start.data_ptr()
            _lib.check(l.azp_nlist_bin(C.byref(a), cursor.data_ptr(), order_tmp.data_ptr(), stream), "azp_nlist_bin")
            return (cell_of, order, cell_start, cursor, order_tmp)
        _lib.check(l.azp_nlist_cell_assign(C.byref(a), stream), "azp_nlist_cell_assign")
        if ncell <= 65536:
            # 16-bit keys: two radix passes instead of four (0.06 instead of 0.16 ms at N = 2^20); same permutation
            k16, order = torch.sort((cell_of - 32768).to(torch.int16), stable=True)
            cell_sorted = k16.to(torch.int32) + 32768
        else:
            cell_sorted, order = torch.sort(cell_of, stable=True)
        order = order.to(torch.int32)
        cell_start = torch.empty(ncell + 1, dtype=torch.int32, device=dev)
        a.d_cell_sorted = cell_sorted.data_ptr()
        a.d_order = order.data_ptr()
        a.d_cell_start = cell_start.data_ptr()
        _lib.check(l.azp_nlist_cell_bounds(C.byref(a), stream), "azp_nlist_cell_bounds")
        return (cell_of, cell_sorted, order, cell_start)

    def rebin_full(self):
        """Half-width cells did not work out for this build (a consumer's plan compile refused them, or somebody needs
        the HOOMD-format list): bin the same positions again into cells of the full list radius."""
        import torch

        a = self._cells
        if a.cell_subdivision != 2:
            return
        a.d_pos = self._pos_at_build.data_ptr()
        dev = self._pos_at_build.device
        bins = self._bin(a, self._box_at_build, self._rl_max, a.n_total, 1, dev, torch.cuda.current_stream(dev).cuda_stream)
        self._keep = (self._keep[0], bins, self._keep[2])

    def _fused_eligible(self):
        if not self.fused or getattr(self, "_compact", False) or len(self._consumers) != 1:
            return False
        c = self._consumers[0]
        return bool(getattr(c, "use_plan", False) and getattr(c, "_planned_entry", None)
                    and getattr(c, "threads_per_particle", 0) in (0, 1) and getattr(c, "use_fused_plan", True))

    def cells_args(self, row_capacity):
        """The binned particles of the last build as azp_nlist_args (fused plan compile). Positions:
        the ones the bins were made from."""
        a = self._cells
        a.d_pos = self._pos_at_build.data_ptr()
        a.row_capacity = int(row_capacity)
        return a

    def leave_fused_mode(self):
        """The consumer could not compile its plan from the cells (unsorted particles, very long
        rows): build the HOOMD-format list for this build after all."""
        if self._fused_active:
            self._fused_active = False
            self._materialize()

    def _materialize(self):
        if self._nlist is None and self.n_neigh is not None:
            import torch

            self.rebin_full()
            a = self._cells
            a.d_pos = self._pos_at_build.data_ptr()
            self._fill(torch.cuda.current_stream(self._pos_at_build.device).cuda_stream)

    def _fill(self, stream):
        """HOOMD-format rows from the binned particles (self._cells)."""
        import torch

        l = _lib.lib()
        a = self._cells
        N = a.N
        n_neigh = self.n_neigh
        dev = n_neigh.device
        done = False
        cap = self._row_capacity
        if self.single_pass and cap > 0 and N and not getattr(self, "_compact", False):
            head = self._head_cache(N, cap, dev)
            size = N * cap
            nlist = torch.empty(size, dtype=torch.int32, device=dev)
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            a.d_head_list = head.data_ptr()
            a.d_nlist = nlist.data_ptr()
            a.row_capacity = cap
            a.d_max_neigh = flag.data_ptr()
            _lib.check(l.azp_nlist_fill(C.byref(a), stream), "azp_nlist_fill")
            stats = torch.stack([n_neigh.max().to(torch.int64), n_neigh.sum(dtype=torch.int64)]).tolist()
            done = stats[0] <= cap  # else: a row overflowed, fall back to exact rows
        if not done:
            a.row_capacity = 0
            a.d_max_neigh = None
            _lib.check(l.azp_nlist_count(C.byref(a), stream), "azp_nlist_count")
            incl = torch.cumsum(n_neigh.to(torch.int64), 0)
            head = incl - n_neigh.to(torch.int64)
            stats = torch.stack([n_neigh.max().to(torch.int64), incl[-1]]).tolist() if N else [0, 0]
            size = int(stats[1])
            nlist = torch.empty(max(size, 1), dtype=torch.int32, device=dev)
            a.d_head_list = head.data_ptr()
            a.d_nlist = nlist.data_ptr()
            _lib.check(l.azp_nlist_fill(C.byref(a), stream), "azp_nlist_fill")
        self._max_neigh, self._n_pairs = int(stats[0]), int(stats[1])
        self._stats_known = True
        # next build: rows with ~6 % head room, multiple of 8 entries (32-B aligned rows)
        self._row_capacity = (int(self._max_neigh * 1.06) + 4 + 7) // 8 * 8
        self._head_list, self._nlist, self._size = head, nlist, size

    def _head_cache(self, N, cap, dev):
        import torch

        key = (N, cap, str(dev))
        if getattr(self, "_head_key", None) != key:
            self._head_fixed = torch.arange(N, dtype=torch.int64, device=dev) * cap
            self._head_key = key
        return self._head_fixed
