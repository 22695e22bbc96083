"""The tile plan compiled straight from the cell list (azp_pair_plan_build_from_cells,
csrc/pair_plan_cells.hip): neighbor search (HOOMD's NeighborListGPUBinned
role for the reference's pair potentials, src/PairPotentials.h:1-60 / hoomd PotentialPairGPU)
and plan compile in one kernel, no HOOMD-format list.

Parity bar: forces, energies and virials through a plan made from the cells == the oracle's
on the oracle's own neighbor list (1e-11 relative, FP64), for multi-type tables, all shift
modes, ghosts + a non-periodic axis, bonded exclusions, boxes of 2-3 cells per axis (rows
wrap inside the tile's cell block), a ragged last tile, particles that moved after the
build (with and without the displacement bound), sub-range launches; and the fallbacks:
rows longer than the capacity (retry with longer rows), rows / staged sets beyond the hard
limits and particles in random memory order (list-based path).
"""
import ctypes as C

import numpy as np
import pytest

import helpers as H
from azplugins_amd import synthetic as syn
from test_gpu_parity import PAIR_PARAMS, _params_table, assert_close

pytestmark = pytest.mark.gpu

_SUB = 1


@pytest.fixture(autouse=True, params=[1, 2], ids=["cells", "halfcells"])
def cell_subdivision(request):
    """Every case twice: cells of the full list radius (27 cells around a member cell) and cells of half of it (the
    5 x 5 x 5 cells around a member's own, cut down per member; azp_nlist_args.cell_subdivision = 2) -- through the C ABI
    (``fused_forces``) and through the API (``nlist.Cell.half_cells``: 0 never, 2 whenever the compiler takes them)."""
    global _SUB
    from azplugins_amd import nlist

    old = nlist.Cell.half_cells
    _SUB = request.param
    nlist.Cell.half_cells = 2 if request.param == 2 else 0
    yield request.param
    nlist.Cell.half_cells = old
    _SUB = 1


PLJ = "PerturbedLennardJones"
PLANNED = {
    PLJ: "azp_pair_forces_planned_perturbed_lennard_jones",
    "Hertz": "azp_pair_forces_planned_hertz",
    "ExpandedYukawa": "azp_pair_forces_planned_expanded_yukawa",
    "Colloid": "azp_pair_forces_planned_colloid",
}


def fused_forces(name, pos, box, params, r_cut, r_buff, ntypes=1, N=None, mode="none", r_on=0.0, virial=False, exclusions=None,
                 row_capacity=0, moved=None, bound=None, prange=None, r_inner=None, balance=False, disp=None, sub=0):
    """Bin, compile the plan from the bins, run the planned kernel; returns (force[, virial], info)."""
    import torch

    from azplugins_amd import _lib

    rc = np.broadcast_to(np.asarray(r_cut, dtype=np.float64), (ntypes, ntypes))
    rl = np.where(rc > 0, rc + r_buff, 0.0)
    cells, keep = H.gpu_cells(pos, box, rl, ntypes, N, exclusions, row_capacity, sub=sub if sub else _SUB)
    n_total = pos.shape[0]
    N = n_total if N is None else N
    dummy = (np.zeros(N, np.uint32), np.zeros(N, np.uint64), np.zeros(1, np.uint32))
    a, t = H.gpu_pair_args(pos, box, dummy, ntypes, r_cut, r_on, mode, virial, N=N, r_list_max=float(rc.max() + 2 * r_buff))
    a.d_n_neigh = keep["n_neigh"].data_ptr()
    if r_inner is not None:
        t["rinnersq"] = H._dev(np.full(ntypes * ntypes, float(r_inner) ** 2))
        a.d_rinnersq = t["rinnersq"].data_ptr()
    plan = _lib.PairPlan()
    plan.set_balance(balance)
    plan.build_from_cells(cells, a, H._stream())
    info = plan.info()
    info["cell_subdivision"] = int(cells.cell_subdivision)
    if not info["valid"] and cells.cell_subdivision == 2 and info["invalid_reason"] in (4, 5):
        # half-width cells refused (the local grid of some tile has more than 2,048 cells): cells of the full radius, as
        # azplugins_amd.pair does
        return fused_forces(name, pos, box, params, r_cut, r_buff, ntypes, N, mode, r_on, virial, exclusions, row_capacity, moved,
                            bound, prange, r_inner, balance, disp, sub=1)
    if not info["valid"]:
        return None, info
    a.d_nlist, a.d_head_list, a.size_nlist = info["list_id"], info["head_id"], 0
    if moved is not None:
        t["pos"].copy_(torch.from_numpy(np.ascontiguousarray(moved)).to("cuda:0"))
    if bound is not None:
        a.has_displacement_bound, a.displacement_bound = 1, float(bound)
    if disp is not None:
        t["disp"] = torch.from_numpy(np.ascontiguousarray(disp, dtype=np.float32)).to("cuda:0")
        a.d_displacement = t["disp"].data_ptr()
    if prange is not None:
        a.range_first, a.range_count = prange
    p = H._dev(np.atleast_2d(params).astype(np.float64))
    _lib.check(getattr(_lib.lib(), PLANNED[name])(plan.handle, C.byref(a), p.data_ptr(), H._stream()), PLANNED[name])
    info["n_neigh"] = keep["n_neigh"].cpu().numpy()
    return H._finish(t, virial), info


@pytest.mark.parametrize("mode", ["none", "shift", "xplor"])
@pytest.mark.parametrize("T", [1, 3])
@pytest.mark.parametrize("name", [PLJ, "Hertz", "ExpandedYukawa", "Colloid"])
def test_fused_plan_parity(oracle, name, T, mode):
    """18^3 sites = 22.8 tiles (ragged last tile), per-type-pair cutoffs, virials."""
    a_lat = 1.1 if name != "Colloid" else 1.6
    pos, L, typeid = H.lattice_config(18, a_lat, 0.1 * a_lat, seed=11, ntypes=T)
    box = oracle.make_box(L)
    r_cut = np.full((T, T), 2.5 if name != "Colloid" else 3.2)
    if T > 1:
        r_cut[0, 1] = r_cut[1, 0] = r_cut[0, 0] - 0.4
        r_cut[2, 2] = r_cut[0, 0] - 0.8
        r_cut[0, 2] = r_cut[2, 0] = 0.0   # a pair of types that does not interact at all
    r_on = 0.8 * r_cut
    params = _params_table(oracle, name, T)
    r_buff = 0.3
    rl = np.where(r_cut > 0, r_cut + r_buff, 0.0)
    nl = oracle.build_nlist(pos, box, rl, ntypes=T)
    f_ref, v_ref = oracle.pair_forces(name, pos, box, nl, params, r_cut, r_on, mode, ntypes=T, virial=True)
    (f_gpu, v_gpu), info = fused_forces(name, pos, (L,), params, r_cut, r_buff, ntypes=T, mode=mode, r_on=r_on, virial=True)
    assert info["valid"] == 1 and info["from_cells"] == 1
    if info["cell_subdivision"] == 1:
        assert 0 < info["max_member_cells"] <= 128  # (what a run watches to re-sort its particles before the compiler refuses)
    assert_close(f_gpu, f_ref)
    assert_close(v_gpu, v_ref, what="virial")
    # the rows are a superset of the exact list by a hair at most (single-precision test, 1e-5 margin)
    extra = info["n_neigh"].astype(np.int64) - nl[0].astype(np.int64)
    assert extra.min() >= 0 and extra.sum() <= 1e-3 * nl[0].sum()
    assert info["max_row"] == info["n_neigh"].max()


def test_fused_plan_ghosts_nonperiodic_and_ranges(oracle):
    """Locals first, ghosts after them (index >= N), x not periodic; whole launch and
    the interior | boundary sub-range launches a decomposed run makes."""
    cfg = syn.config_plj_sc(12)
    xyz, L = cfg["xyz"], cfg["L"]
    order = np.argsort(xyz[:, 0] > 0.0, kind="stable")
    pos = syn.pos4(xyz[order])
    N = int((xyz[:, 0] <= 0.0).sum())
    box_o = oracle.make_box(L, periodic=(0, 1, 1))
    params = oracle.pack_pair_params(PLJ, cfg["params"])
    nl = oracle.build_nlist(pos, box_o, 2.9, N=N)
    assert nl[2].max() >= N
    f_ref = oracle.pair_forces(PLJ, pos, box_o, nl, params, 2.5, N=N)
    boxg = (L, (0, 0, 0), (0, 1, 1))
    f_gpu, info = fused_forces(PLJ, pos, boxg, params, 2.5, 0.4, N=N)
    assert info["valid"] == 1 and info["from_cells"] == 1
    assert f_gpu.shape == (N, 4)
    assert_close(f_gpu, f_ref)
    first = 300
    for rng in ((0, first), (first, N - first)):
        f_part, _ = fused_forces(PLJ, pos, boxg, params, 2.5, 0.4, N=N, prange=rng)
        sl = slice(rng[0], rng[0] + rng[1])
        assert_close(f_part[sl], f_ref[sl])
        # planned kernels round the range outwards to whole tiles of 256 (include/azp.h): rows of
        # tiles outside it are not touched, rows they do write are right
        lo, hi = rng[0] // 256 * 256, min(-(-(rng[0] + rng[1]) // 256) * 256, N)
        rest = np.ones(N, bool)
        rest[lo:hi] = False
        assert np.isnan(f_part[rest]).all()
        assert_close(f_part[lo:hi], f_ref[lo:hi])


def test_fused_plan_exclusions(oracle):
    """Bonded partners are left out of the rows (HOOMD's exclusions=('bond',))."""
    cfg = syn.config_chains(16, 16, 8, 8)
    n = cfg["xyz"].shape[0]
    pos = syn.pos4(cfg["xyz"])
    n_excl = np.zeros(n, dtype=np.uint32)
    excl = np.zeros((n, 2), dtype=np.uint32)
    for a_, b_ in cfg["bonds"]:
        for me, other in ((a_, b_), (b_, a_)):
            excl[me, n_excl[me]] = other
            n_excl[me] += 1
    box = oracle.make_box(cfg["L"])
    params = oracle.pack_pair_params("Hertz", dict(epsilon=2.0))
    nl = oracle.build_nlist(pos, box, 2.3, exclusions=(n_excl, excl))
    nl_all = oracle.build_nlist(pos, box, 2.3)
    assert nl_all[0].sum() > nl[0].sum()
    f_ref = oracle.pair_forces("Hertz", pos, box, nl, params, 2.0)
    f_gpu, info = fused_forces("Hertz", pos, (cfg["L"],), params, 2.0, 0.3, exclusions=(n_excl, excl))
    assert info["valid"] == 1 and info["from_cells"] == 1
    assert_close(f_gpu, f_ref)
    assert np.array_equal(info["n_neigh"] >= nl[0], np.ones(n, bool)) and info["n_neigh"].sum() < nl_all[0].sum()


@pytest.mark.parametrize("n_side,r_cut", [(6, 2.5), (8, 2.5), (7, 1.2), (5, 2.0)])
def test_fused_plan_small_boxes(oracle, n_side, r_cut):
    """2 - 3 cells per axis: every tile's cell block covers the whole periodic axis, neighbor
    cells wrap inside the block and pairs need the per-pair minimum image."""
    pos, L, typeid = H.lattice_config(n_side, 1.1, 0.11, seed=5, ntypes=2)
    r_buff = 0.25
    assert L[0] >= 2 * (r_cut + r_buff)
    box = oracle.make_box(L)
    tab = H.sym_table(2, PAIR_PARAMS[PLJ])
    params = np.array([oracle.pack_pair_params(PLJ, tab[i][j]) for i in range(2) for j in range(2)])
    nl = oracle.build_nlist(pos, box, r_cut + r_buff, ntypes=2)
    f_ref = oracle.pair_forces(PLJ, pos, box, nl, params, r_cut, 0.0, "shift", ntypes=2)
    f_gpu, info = fused_forces(PLJ, pos, (L,), params, r_cut, r_buff, ntypes=2, mode="shift")
    assert info["valid"] == 1 and info["from_cells"] == 1
    assert_close(f_gpu, f_ref)
    assert info["n_neigh"].sum() >= nl[0].sum()


@pytest.mark.parametrize("frac", [0.0, 0.3, 0.99])
def test_fused_plan_after_particles_moved(oracle, frac):
    """Plan compiled at the build positions, kernel run after every particle moved by up to
    frac * r_buff / 2 (some through the periodic boundary): with the bound and without it."""
    cfg = syn.config_plj_sc(20)
    n = cfg["xyz"].shape[0]
    L = cfg["L"]
    cfg["xyz"] = syn.wrap(cfg["xyz"] + np.array([0.42 * 0.8 ** (-1.0 / 3.0), 0.0, 0.0]), L)
    pos0 = syn.pos4(cfg["xyz"])
    box = oracle.make_box(L)
    r_cut, r_buff = 3.0, 0.4
    params = oracle.pack_pair_params(PLJ, cfg["params"])
    nl = oracle.build_nlist(pos0, box, r_cut + r_buff)
    tag = np.arange(n, dtype=np.uint64)
    v = np.stack([syn.normal(77, tag, c) for c in range(3)], axis=1)
    v *= (frac * 0.25 * r_buff * syn.u01(78, tag, 0) / np.linalg.norm(v, axis=1))[:, None]
    v[:, 0] += frac * 0.25 * r_buff
    amp = np.linalg.norm(v, axis=1).max()
    assert amp <= 0.5 * r_buff
    moved = syn.pos4(syn.wrap(cfg["xyz"] + v, L))
    f_ref = oracle.pair_forces(PLJ, moved, box, nl, params, r_cut, 0.0, "shift", nthreads=8)
    r_wca = 2.0 ** (1.0 / 6.0) * cfg["params"]["sigma"]
    for bound in (None, amp * (1 + 1e-12)):
        f_gpu, info = fused_forces(PLJ, pos0, (L,), params, r_cut, r_buff, mode="shift", moved=moved, bound=bound, r_inner=r_wca + r_buff)
        assert info["valid"] == 1
        assert_close(f_gpu, f_ref)


@pytest.mark.parametrize("few", [0.002, 0.05, 1.0])
def test_fused_plan_local_displacement_bound(oracle, few):
    """Per-particle displacements (azp_pair_args.d_displacement): a fraction ``few`` of the particles moves by up to
    0.99 r_buff / 2, the rest by a hundredth of that, so most tiles stop their rows many shells before what the global
    bound dictates -- and the forces are still the oracle's on the moved positions with the old list. The inner radius
    is the one azplugins_amd.pair passes (r_wca + r_buff + 1e-3: row phases active)."""
    cfg = syn.config_plj_sc(20)
    n = cfg["xyz"].shape[0]
    L = cfg["L"]
    pos0 = syn.pos4(cfg["xyz"])
    box = oracle.make_box(L)
    r_cut, r_buff = 3.0, 0.4
    params = oracle.pack_pair_params(PLJ, cfg["params"])
    nl = oracle.build_nlist(pos0, box, r_cut + r_buff)
    tag = np.arange(n, dtype=np.uint64)
    v = np.stack([syn.normal(91, tag, c) for c in range(3)], axis=1)
    v /= np.linalg.norm(v, axis=1)[:, None]
    fast = syn.u01(92, tag, 0) < few
    length = np.where(fast, 0.99, 0.0099) * 0.5 * r_buff * syn.u01(93, tag, 1)
    v *= length[:, None]
    moved = syn.pos4(syn.wrap(cfg["xyz"] + v, L))
    f_ref = oracle.pair_forces(PLJ, moved, box, nl, params, r_cut, 0.0, "shift", nthreads=8)
    r_wca = 2.0 ** (1.0 / 6.0) * cfg["params"]["sigma"]
    disp = np.nextafter(np.linalg.norm(v, axis=1).astype(np.float32) * np.float32(1.000001), np.float32(np.inf))
    from azplugins_amd import _lib

    results = []
    for phases in (1, 0):  # the row phases of the tile kernel (off by default) on and off
        old = _lib.lib().azp_tuning_set(1, phases)
        try:
            f_gpu, info = fused_forces(PLJ, pos0, (L,), params, r_cut, r_buff, mode="shift", moved=moved, bound=float(disp.max()), disp=disp,
                                       r_inner=r_wca + r_buff + 1e-3)
            assert info["valid"] == 1 and info["core_radius"] > 0 and info["sure_radius"] > 0
            assert_close(f_gpu, f_ref)
            # the same launch with the global bound alone gives the same bits (both are exact evaluations of the same
            # pairs in the same order; only the number of skipped out-of-range entries differs)
            f_glob, _ = fused_forces(PLJ, pos0, (L,), params, r_cut, r_buff, mode="shift", moved=moved, bound=float(disp.max()),
                                     r_inner=r_wca + r_buff + 1e-3)
            assert np.array_equal(f_gpu, f_glob)
            results.append(f_gpu)
        finally:
            _lib.lib().azp_tuning_set(1, old)
    assert_close(results[0], results[1])


def test_fused_plan_row_capacity_protocol(oracle):
    """A row longer than the capacity invalidates the plan with reason 3 and reports the
    longest row; the caller retries with longer rows (HOOMD's protocol for its own list)."""
    cfg = syn.config_plj_sc(16)
    pos = syn.pos4(cfg["xyz"])
    L = cfg["L"]
    params = oracle.pack_pair_params(PLJ, cfg["params"])
    out, info = fused_forces(PLJ, pos, (L,), params, 3.0, 0.4, row_capacity=64)
    assert out is None and info["valid"] == 0 and info["invalid_reason"] == 3 and info["from_cells"] == 1
    box = oracle.make_box(L)
    nl = oracle.build_nlist(pos, box, 3.4)
    assert info["max_row"] >= nl[0].max() > 64
    cap = (info["max_row"] + 7) // 8 * 8
    f_gpu, info = fused_forces(PLJ, pos, (L,), params, 3.0, 0.4, row_capacity=cap)
    assert info["valid"] == 1 and info["row_capacity"] == cap
    assert_close(f_gpu, oracle.pair_forces(PLJ, pos, box, nl, params, 3.0))


def test_fused_plan_limits_fall_back(oracle):
    """Unsorted particles (a tile's cells are all over the box) and rows beyond the hard
    limit are reported, never mis-computed; through the API the list-based path takes over
    and the forces still match the oracle."""
    import azplugins_amd as azp

    cfg = syn.config_plj_sc(20)
    n = cfg["xyz"].shape[0]
    L = cfg["L"]
    perm = np.argsort(syn.hash64(5, np.arange(n, dtype=np.uint64), 0), kind="stable")
    xyz = cfg["xyz"][perm]
    pos = syn.pos4(xyz)
    params = oracle.pack_pair_params(PLJ, cfg["params"])
    out, info = fused_forces(PLJ, pos, (L,), params, 2.5, 0.4)
    assert out is None and info["valid"] == 0 and info["invalid_reason"] in (2, 4, 5)

    box = oracle.make_box(L)
    for r_cut, r_buff, reason in ((2.5, 0.4, (4, 5)), (5.1, 0.3, (2, 3))):
        x = xyz if reason == (4, 5) else cfg["xyz"]
        sim = azp.Simulation(device="cuda:0", seed=1)
        sim.create_state_from_snapshot(azp.Snapshot.from_arrays(x, L))
        nl = azp.nlist.Cell(buffer=r_buff)
        pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=r_cut)
        pot.params[("A", "A")] = cfg["params"]
        sim.operations.integrator = azp.Integrator(dt=0.001, forces=[pot])
        sim.operations.tuners.clear()   # no particle sorter: the memory order stays random
        sim.run(0)
        assert not nl._fused_active and nl._nlist is not None
        onl = oracle.build_nlist(syn.pos4(x), box, r_cut + r_buff)
        f_ref = oracle.pair_forces(PLJ, syn.pos4(x), box, onl, params, r_cut, nthreads=8)
        assert_close(np.c_[pot.forces, pot.energies], f_ref)
        assert nl.n_pairs == int(onl[0].sum())


def test_unsorted_tiles_ask_for_a_particle_sort(oracle):
    """Particles in random memory order with a sorter whose period is far away: the plan cannot be compiled from the cells
    (the members of a tile are all over the box), the list-based path computes that step, and the run sorts the particles
    at its next step instead of paying list-based rebuilds until the sorter's period comes round -- after which the plan
    comes from the cells again (a DPD fluid gets there by diffusion between two sorts)."""
    import azplugins_amd as azp

    cfg = syn.config_plj_sc(20)
    n = cfg["xyz"].shape[0]
    L = cfg["L"]
    perm = np.argsort(syn.hash64(6, np.arange(n, dtype=np.uint64), 0), kind="stable")
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"][perm], L, tag=perm.astype(np.uint32)))
    nl = azp.nlist.Cell(buffer=0.4)
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=2.5)
    pot.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=0.0005, forces=[pot], methods=[azp.ConstantVolume()])
    sim.operations.tuners.clear()
    sorter = azp.ParticleSorter(trigger_period=100000)
    sim.operations.tuners.append(sorter)
    sim.run(0)
    assert not nl._fused_active and nl._sort_wanted and sorter.num_sorts == 0
    sim.run(3)
    assert sorter.num_sorts == 1 and nl.fused and nl._fused_active and not nl._sort_wanted
    assert pot.plan_info["valid"] == 1 and pot.plan_info["from_cells"] == 1
    x = syn.pos4(sim.state.pos[:, :3].cpu().numpy())
    box = oracle.make_box(L)
    onl = oracle.build_nlist(x, box, 2.9)
    f_ref = oracle.pair_forces(PLJ, x, box, onl, oracle.pack_pair_params(PLJ, cfg["params"]), 2.5, nthreads=8)
    assert_close(np.c_[pot.forces, pot.energies], f_ref)
    # the tags travelled with the particles
    assert np.array_equal(np.sort(sim.state.tag.cpu().numpy().view(np.uint32)), np.arange(n, dtype=np.uint32))


def test_fused_plan_through_the_api(oracle):
    """hoomd.azplugins-shaped run: the sole consumer of a Cell list gets its plan from the
    cells at every rebuild (no u32 list is ever filled), forces during an NVE run match the
    oracle at the current positions, list statistics and the HOOMD-format arrays appear on
    demand, switching the tile path off materializes the list."""
    import azplugins_amd as azp

    cfg = syn.config_plj_sc(20)
    L = cfg["L"]
    box = oracle.make_box(L)
    params = oracle.pack_pair_params(PLJ, cfg["params"])
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], L))
    nl = azp.nlist.Cell(buffer=0.4)
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=3.0, mode="shift")
    pot.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot], methods=[azp.ConstantVolume()])
    sim.run(0)
    sim.thermalize_particle_momenta(1.0, seed=3)
    assert nl._fused_active and nl._nlist is None
    assert pot.plan_info["valid"] == 1 and pot.plan_info["from_cells"] == 1
    builds = nl.num_builds
    checked = 0
    for step in range(30):
        sim.run(1)
        if step % 7 == 0 or nl.num_builds != builds:
            builds = nl.num_builds
            pos = syn.pos4(sim.state.pos[: sim.state.N, :3].cpu().numpy())
            onl = oracle.build_nlist(pos, box, 3.0)
            f_ref = oracle.pair_forces(PLJ, pos, box, onl, params, 3.0, 0.0, "shift", nthreads=8)
            assert_close(np.c_[pot.forces, pot.energies], f_ref)
            checked += 1
    assert nl.num_builds >= 3 and checked >= 5
    assert nl._nlist is None                      # 30 steps, several rebuilds, no u32 list
    assert pot.plan_info["from_cells"] == 1
    # mode change in the middle of a list's life: the plan is recompiled from the bins of the build
    pot.mode = "none"
    pot.compute(0)
    pos = syn.pos4(sim.state.pos[: sim.state.N, :3].cpu().numpy())
    onl = oracle.build_nlist(pos, box, 3.0)
    assert_close(np.c_[pot.forces, pot.energies], oracle.pair_forces(PLJ, pos, box, onl, params, 3.0, nthreads=8))
    # statistics and arrays on demand
    n_pairs = nl.n_pairs
    assert nl._nlist is None and n_pairs > 0
    listed = int(nl.n_neigh.sum().item())
    head, rows = nl.head_list, nl.nlist           # materializes the HOOMD-format list for this build
    assert nl._nlist is not None and rows.numel() >= nl.n_pairs
    assert 0 <= listed - nl.n_pairs <= 1e-3 * listed
    f_tile = pot.force_tensor.clone()
    pot.use_plan = False                          # generic kernel on the u32 list
    pot.compute(0)
    assert_close(pot.force_tensor.cpu().numpy(), f_tile.cpu().numpy())
    pot.use_plan = True
    sim.run(10)
    pos = syn.pos4(sim.state.pos[: sim.state.N, :3].cpu().numpy())
    onl = oracle.build_nlist(pos, box, 3.0)
    assert_close(np.c_[pot.forces, pot.energies], oracle.pair_forces(PLJ, pos, box, onl, params, 3.0, nthreads=8))


def test_two_consumers_share_a_real_list(oracle):
    """Two pair potentials on one neighbor list: the list is built in HOOMD's format (the
    plan-from-cells shortcut is for a list with a single tile-kernel consumer), both get their
    plans from it, each force matches the oracle with its own cutoff."""
    import azplugins_amd as azp

    pos, L, typeid = H.lattice_config(14, 1.1, 0.11, seed=3, ntypes=1)
    box = oracle.make_box(L)
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(pos[:, :3], L))
    nl = azp.nlist.Cell(buffer=0.3)
    plj = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=2.5, mode="shift")
    plj.params[("A", "A")] = PAIR_PARAMS[PLJ](0, 0)
    hz = azp.pair.Hertz(nlist=nl, default_r_cut=1.4)
    hz.params[("A", "A")] = dict(epsilon=3.0)
    sim.operations.integrator = azp.Integrator(dt=0.001, forces=[plj, hz])
    sim.run(0)
    assert not nl._fused_active and nl._nlist is not None
    assert plj.plan_info["from_cells"] == 0 and hz.plan_info["from_cells"] == 0
    onl = oracle.build_nlist(pos, box, 2.8)
    f1 = oracle.pair_forces(PLJ, pos, box, onl, oracle.pack_pair_params(PLJ, PAIR_PARAMS[PLJ](0, 0)), 2.5, 0.0, "shift")
    f2 = oracle.pair_forces("Hertz", pos, box, onl, oracle.pack_pair_params("Hertz", dict(epsilon=3.0)), 1.4)
    assert_close(np.c_[plj.forces, plj.energies], f1)
    assert_close(np.c_[hz.forces, hz.energies], f2)


def test_balanced_plan_is_the_same_list(oracle):
    """azp_pair_plan_set_balance: rows go to the force kernel's lanes longest in-range row first.
    Same pairs, same per-particle sums up to the order of the additions inside a class; ragged
    rows (a random fluid at low density), a ragged last tile, sub-range launches."""
    cfg = syn.config_dpd(5000)           # uniform random positions, rho = 3: rows of 5 .. 25 in-range entries
    pos = syn.pos4(cfg["xyz"])
    L = cfg["L"]
    box = oracle.make_box(L)
    params = oracle.pack_pair_params("Hertz", dict(epsilon=2.5))
    nl = oracle.build_nlist(pos, box, 1.4)
    f_ref, v_ref = oracle.pair_forces("Hertz", pos, box, nl, params, 1.0, virial=True)
    (f_plain, v_plain), info0 = fused_forces("Hertz", pos, (L,), params, 1.0, 0.4, virial=True)
    (f_bal, v_bal), info1 = fused_forces("Hertz", pos, (L,), params, 1.0, 0.4, virial=True, balance=True)
    assert info0["balanced"] == 0 and info1["balanced"] == 1 and info1["valid"] == 1
    assert_close(f_plain, f_ref)
    assert_close(f_bal, f_ref)
    assert_close(v_bal, v_ref, what="virial")
    n = pos.shape[0]
    f_part, _ = fused_forces("Hertz", pos, (L,), params, 1.0, 0.4, balance=True, prange=(512, 1024))
    assert_close(f_part[512:1536], f_ref[512:1536])
    assert np.isnan(f_part[:512]).all() and np.isnan(f_part[1536:]).all()
    assert n % 256 != 0


@pytest.mark.parametrize("kind", ["dpd", "tpm"])
def test_balanced_plans_through_the_api(oracle, kind):
    """The DPD thermostat and TwoPatchMorse ask for balanced plans (their pair blocks are expensive,
    their rows short and ragged): forces (and torques) against the oracle, during a short NVE run."""
    import azplugins_amd as azp

    if kind == "dpd":
        cfg = syn.config_dpd(6000)
        sim = azp.Simulation(device="cuda:0", seed=cfg["seed"])
        sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], velocity=cfg["vel"], tag=cfg["tag"]))
        nl = azp.nlist.Cell(buffer=cfg["r_buff"])
        pot = azp.pair.DPDGeneralWeight(nlist=nl, kT=cfg["kT"], default_r_cut=cfg["r_cut"])
        dt = cfg["dt"]
    else:
        cfg = syn.config_tpm(12, 12, 16)
        sim = azp.Simulation(device="cuda:0", seed=1)
        sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], orientation=cfg["orientation"]))
        nl = azp.nlist.Cell(buffer=cfg["r_buff"])
        pot = azp.pair.TwoPatchMorse(nlist=nl, default_r_cut=cfg["r_cut"], mode="shift")
        dt = 0.002
    pot.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=dt, forces=[pot], methods=[azp.ConstantVolume()])
    sim.operations.tuners.clear()
    sim.run(0)
    info = pot.plan_info
    assert info["valid"] == 1 and info["from_cells"] == 1 and info["balanced"] == 1
    box = oracle.make_box(cfg["L"])
    for steps in (0, 6):
        sim.run(steps)
        pot.compute(sim.timestep)   # (the run's last evaluation saw the half-kicked velocities)
        st = sim.state
        n = st.N
        pos = st.pos[:n].cpu().numpy()
        onl = oracle.build_nlist(pos, box, cfg["r_cut"])
        if kind == "dpd":
            vel = st.vel[:n].cpu().numpy()
            tag = st.tag[:n].cpu().numpy().view(np.uint32)
            params = oracle.pack_pair_params("DPDGeneralWeight", cfg["params"])
            f_ref = oracle.dpd_forces(pos, vel, tag, box, onl, params, cfg["r_cut"], kT=cfg["kT"], dt=dt, seed=cfg["seed"],
                                      timestep=sim.timestep)
            assert_close(np.c_[pot.forces, pot.energies], f_ref, what="dpd after %d steps" % steps)
        else:
            q = st.orientation[:n].cpu().numpy()
            params = oracle.pack_pair_params("TwoPatchMorse", cfg["params"])
            f_ref, t_ref = oracle.aniso_forces_tpm(pos, q, box, onl, params, cfg["r_cut"], "shift")
            assert_close(np.c_[pot.forces, pot.energies], f_ref, what="tpm force after %d steps" % steps)
            assert_close(pot.torques, t_ref[:, :3], what="tpm torque")


def _random_case(seed):
    """A random small system for the plan-from-cells compiler: box shape, periodicity, density, number
    of types, per-pair cutoffs, buffer, N not a multiple of 256, memory order by cells."""
    rng = np.random.RandomState(seed)
    ntypes = int(rng.randint(1, 4))
    r_cut_max = float(rng.uniform(1.0, 3.0))
    r_buff = float(rng.uniform(0.1, 0.5))
    r_list = r_cut_max + r_buff
    cells = rng.randint(2, 7, size=3)                       # cells per axis
    L = cells * r_list * rng.uniform(1.0, 1.3, size=3)
    rho = float(rng.uniform(0.15, 0.9)) * min(1.0, (2.6 / r_list) ** 3 * 1.2)   # keep rows and staged sets within the limits
    n = int(max(40, min(9000, rho * np.prod(L))))
    periodic = tuple(int(v) for v in rng.randint(0, 2, size=3))
    if rng.rand() < 0.5:
        periodic = (1, 1, 1)
    # jittered lattice positions (no overlapping pairs: the potentials stay finite), then a cell-curve memory order
    side = int(np.ceil(n ** (1.0 / 3.0)))
    grid = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), axis=-1).reshape(-1, 3)[:n]
    a = L / side
    tag = np.arange(n, dtype=np.uint64)
    jit = np.stack([syn.u01(seed * 7 + 1, tag, c) - 0.5 for c in range(3)], axis=1) * 0.6
    xyz = (grid + 0.5 + jit) * a - 0.5 * L
    xyz = np.clip(xyz, -0.5 * L * (1 - 1e-9), 0.5 * L * (1 - 1e-9))
    dim = np.maximum((L / (0.5 * r_list)).astype(np.int64), 1)
    c = np.minimum(((xyz + 0.5 * L) / (L / dim)).astype(np.int64), dim - 1)
    b = 2
    key = ((c[:, 2] // b) * ((dim[1] + b - 1) // b) + (c[:, 1] // b)) * ((dim[0] + b - 1) // b) + (c[:, 0] // b)
    inner = ((c[:, 2] % b) * b + (c[:, 1] % b)) * b + (c[:, 0] % b)
    order = np.lexsort((inner, key))
    xyz = xyz[order]
    typeid = (syn.hash64(seed + 5, tag, 3) % np.uint64(ntypes)).astype(np.int64)
    r_cut = np.full((ntypes, ntypes), r_cut_max)
    for i in range(ntypes):
        for j in range(i, ntypes):
            r_cut[i, j] = r_cut[j, i] = r_cut_max * float(rng.choice([1.0, 0.8, 0.6]))
    r_cut[0, 0] = r_cut_max
    min_sep = 0.4 * float(a.min())
    return dict(pos=syn.pos4(xyz, typeid), L=L, periodic=periodic, ntypes=ntypes, r_cut=r_cut, r_buff=r_buff, min_sep=min_sep,
                mode=str(rng.choice(["none", "shift", "xplor"])), name=str(rng.choice([PLJ, "Hertz", "ExpandedYukawa"])),
                n_ghost=int(rng.randint(0, n // 4)) if periodic != (1, 1, 1) else 0, balance=bool(rng.randint(0, 2)))


@pytest.mark.parametrize("seed", range(40))
def test_fused_plan_random_systems(oracle, seed):
    """Forty random systems (box shape 2-6 cells per axis, any periodicity, densities 0.15-0.9, 1-3 types with
    per-pair cutoffs, ghosts, balanced or not): either the plan from the cells gives the oracle's forces, or it
    is reported invalid with one of the documented reasons."""
    cse = _random_case(seed)
    pos, L, T = cse["pos"], cse["L"], cse["ntypes"]
    N = pos.shape[0] - cse["n_ghost"]
    box_o = oracle.make_box(L, periodic=cse["periodic"])
    rl = cse["r_cut"] + cse["r_buff"]
    nl = oracle.build_nlist(pos, box_o, rl, N=N, ntypes=T)
    tab = H.sym_table(T, PAIR_PARAMS[cse["name"]])
    params = np.array([oracle.pack_pair_params(cse["name"], tab[i][j]) for i in range(T) for j in range(T)])
    r_on = 0.8 * cse["r_cut"]
    f_ref, v_ref = oracle.pair_forces(cse["name"], pos, box_o, nl, params, cse["r_cut"], r_on, cse["mode"], N=N, ntypes=T, virial=True)
    out, info = fused_forces(cse["name"], pos, (L, (0, 0, 0), cse["periodic"]), params, cse["r_cut"], cse["r_buff"], ntypes=T, N=N,
                             mode=cse["mode"], r_on=r_on, virial=True, balance=cse["balance"], row_capacity=504)
    if out is None:
        assert info["invalid_reason"] in (2, 3, 4, 5), info
        return
    assert info["valid"] == 1 and info["from_cells"] == 1 and info["balanced"] == int(cse["balance"])
    assert_close(out[0], f_ref, what="seed %d force" % seed)
    assert_close(out[1], v_ref, what="seed %d virial" % seed)
    extra = info["n_neigh"].astype(np.int64) - nl[0].astype(np.int64)
    assert extra.min() >= 0 and extra.sum() <= 1e-3 * max(nl[0].sum(), 1000)


def test_fused_plan_many_types(oracle):
    """Ten particle types: the compiler reads its per-type-pair tables from memory (they are cached in
    LDS only up to eight types) and classes come from the formula, not from the r^2 table."""
    T = 10
    pos, L, typeid = H.lattice_config(15, 1.1, 0.11, seed=23, ntypes=T)
    box = oracle.make_box(L)
    r_cut = np.full((T, T), 2.5)
    for i in range(T):
        for j in range(i, T):
            r_cut[i, j] = r_cut[j, i] = 2.5 - 0.07 * ((i * 3 + j) % 8)
    tab = H.sym_table(T, lambda i, j: dict(epsilon=1.0 + 0.05 * (i + j), sigma=1.0 - 0.01 * (i + j), attraction_scale_factor=0.1 + 0.04 * min(i, j)))
    params = np.array([oracle.pack_pair_params(PLJ, tab[i][j]) for i in range(T) for j in range(T)])
    nl = oracle.build_nlist(pos, box, r_cut + 0.3, ntypes=T)
    f_ref, v_ref = oracle.pair_forces(PLJ, pos, box, nl, params, r_cut, 0.0, "shift", ntypes=T, virial=True)
    for balance in (False, True):
        (f_gpu, v_gpu), info = fused_forces(PLJ, pos, (L,), params, r_cut, 0.3, ntypes=T, mode="shift", virial=True, balance=balance)
        assert info["valid"] == 1 and info["from_cells"] == 1
        assert_close(f_gpu, f_ref)
        assert_close(v_gpu, v_ref, what="virial")


def test_native_binning_is_the_stable_sort(oracle):
    """azp_nlist_bin (counting sort + per-cell sort) gives exactly the permutation and the cell bounds of a stable
    sort of the particles by cell -- ragged cells, empty cells, ghosts, a cell that holds many particles."""
    import torch

    from azplugins_amd import _lib

    rng = np.random.default_rng(5)
    n = 30000
    L = np.array([20.0, 14.0, 9.0])
    xyz = (rng.random((n, 3)) - 0.5) * L
    xyz[:3000] = 0.3 * (rng.random((3000, 3)) - 0.5) + np.array([3.0, 2.0, -1.0])  # a dense clump: one crowded cell
    pos = syn.pos4(xyz)
    rl = np.array([[1.7]])
    cells, keep = H.gpu_cells(pos, (L,), rl, 1, None, None, 0)
    ncell = int(cells.grid.dim[0] * cells.grid.dim[1] * cells.grid.dim[2])
    cell_of = keep["cell_of"].cpu().numpy().view(np.uint32) if "cell_of" in keep else None
    order = torch.empty(n, dtype=torch.int32, device="cuda:0")
    start = torch.empty(ncell + 1, dtype=torch.int32, device="cuda:0")
    cof = torch.empty(n, dtype=torch.int32, device="cuda:0")
    cursor = torch.empty(ncell, dtype=torch.int32, device="cuda:0")
    tmp = torch.empty(n, dtype=torch.int32, device="cuda:0")
    cells.d_cell_of, cells.d_order, cells.d_cell_start = cof.data_ptr(), order.data_ptr(), start.data_ptr()
    _lib.check(_lib.lib().azp_nlist_bin(C.byref(cells), cursor.data_ptr(), tmp.data_ptr(), H._stream()), "azp_nlist_bin")
    torch.cuda.synchronize()
    c = cof.cpu().numpy()
    ref_order = np.argsort(c, kind="stable")
    assert np.array_equal(order.cpu().numpy(), ref_order)
    ref_start = np.searchsorted(c[ref_order], np.arange(ncell + 1), side="left")
    assert np.array_equal(start.cpu().numpy(), ref_start)
    if cell_of is not None:
        assert np.array_equal(c.view(np.uint32), cell_of)
    assert np.bincount(c, minlength=ncell).max() > 500  # the crowded cell was really there


def test_neighbor_list_buffer_tuner(oracle):
    """azplugins_amd.tune.NeighborListBuffer (hoomd.md.tune.NeighborListBuffer reduced): sweeps the buffer through short
    stretches of the run, leaves the fastest one set -- and the forces on the state it leaves behind are the oracle's."""
    import azplugins_amd as azp

    cfg = syn.config_plj_sc(12)
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"]))
    nl = azp.nlist.Cell(buffer=0.4)
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=2.5, mode="shift")
    pot.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=0.004, forces=[pot], methods=[azp.ConstantVolume()])
    sim.operations.tuners.clear()
    sim.run(0)
    sim.thermalize_particle_momenta(1.0, seed=3)
    tuner = azp.tune.NeighborListBuffer(nl, candidates=(0.3, 0.5, 0.7), steps=12)
    best = tuner.tune(sim)
    assert best in (0.3, 0.5, 0.7) and nl.buffer == best and set(tuner.results) == {0.3, 0.5, 0.7}
    sim.run(5)
    n = sim.state.N
    pos = sim.state.pos[:n].cpu().numpy()
    box = oracle.make_box(cfg["L"])
    onl = oracle.build_nlist(pos, box, 2.5)
    f_ref = oracle.pair_forces(PLJ, pos, box, onl, oracle.pack_pair_params(PLJ, cfg["params"]), 2.5, 0.0, "shift")
    assert_close(np.c_[pot.forces, pot.energies], f_ref)
