"""ParticleSorter (row N1, "+ particle SFC sort"): a shuffled system cannot be tiled
(the planned entry point falls back to the generic kernel); after sorting it can,
and forces / energies are the same per TAG. Bonds are re-indexed."""

import numpy as np
import pytest

import azplugins_amd as azp
from azplugins_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def _by_tag(sim, force):
    tag = sim.state.tag.cpu().numpy().view(np.uint32)[: sim.state.N]
    out = np.zeros((sim.state.N, 4))
    out[tag] = np.c_[force.forces, force.energies]
    return out


def test_sorter_restores_tiling_and_keeps_forces(oracle):
    cfg = syn.config_chains(32, 32, 16, 16)
    n = cfg["xyz"].shape[0]
    perm = np.argsort(syn.u01(41, np.arange(n, dtype=np.uint64), 0))  # deterministic shuffle
    inv = np.empty(n, dtype=np.int64)
    inv[perm] = np.arange(n)
    xyz = cfg["xyz"][perm]
    bonds = inv[np.asarray(cfg["bonds"], dtype=np.int64)]
    snap = azp.Snapshot.from_arrays(xyz, cfg["L"], bonds=bonds, bond_types=("A-A",))
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(snap)
    nl = azp.nlist.Cell(buffer=0.4)
    plj = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=3.0, mode="shift")
    plj.params[("A", "A")] = cfg["params"]
    dw = azp.bond.DoubleWell()
    dw.params["A-A"] = cfg["bond_params"]
    sim.operations.integrator = azp.Integrator(dt=0.001, forces=[plj, dw])
    sim.run(0)
    assert plj.plan_info["valid"] == 0  # random order: a tile would have to stage far too many particles
    f_pair0, f_bond0 = _by_tag(sim, plj), _by_tag(sim, dw)

    sorter = azp.ParticleSorter(trigger_period=0)
    order = sorter.sort(sim).cpu().numpy()
    assert sorted(order.tolist()) == list(range(n))
    sim.run(0)
    assert plj.plan_info["valid"] == 1 and nl.num_builds == 2
    f_pair1, f_bond1 = _by_tag(sim, plj), _by_tag(sim, dw)
    scale = np.abs(f_pair0).max()
    assert np.abs(f_pair1 - f_pair0).max() <= 1e-10 * scale
    assert np.abs(f_bond1 - f_bond0).max() <= 1e-12 * max(np.abs(f_bond0).max(), 1.0)
    # and against the oracle on the original (unshuffled) arrays; tags of the shuffled snapshot are 0..n-1 in
    # shuffled order, so map back through perm
    pos = syn.pos4(cfg["xyz"])
    box = oracle.make_box(cfg["L"])
    excl_n = np.zeros(n, dtype=np.uint32)
    excl = np.zeros((n, 2), dtype=np.uint32)
    for a_, b_ in np.asarray(cfg["bonds"], dtype=np.int64):
        for me, other in ((a_, b_), (b_, a_)):
            excl[me, excl_n[me]] = other
            excl_n[me] += 1
    o_nl = oracle.build_nlist(pos, box, 3.4, exclusions=(excl_n, excl))
    ref = oracle.pair_forces("PerturbedLennardJones", pos, box, o_nl,
                             oracle.pack_pair_params("PerturbedLennardJones", cfg["params"]), 3.0, mode="shift")
    got = np.zeros_like(ref)
    got[perm] = f_pair1  # tag t of the shuffled snapshot is original particle perm[t]
    assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max()


def test_sorter_in_run_loop_keeps_trajectory():
    """Sorting every 5 steps during an NVE run changes nothing per tag."""
    cfg = syn.config_plj_sc(12)
    n = cfg["xyz"].shape[0]
    tag = np.arange(n, dtype=np.uint64)
    vel = np.stack([syn.normal(9, tag, c) for c in range(3)], axis=1) * 0.7
    vel -= vel.mean(axis=0)
    out = []
    for period in (0, 5):
        sim = azp.Simulation(device="cuda:0", seed=1)
        sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], velocity=vel))
        nl = azp.nlist.Cell(buffer=0.4)
        pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=2.5, mode="shift")
        pot.params[("A", "A")] = cfg["params"]
        sim.operations.integrator = azp.Integrator(dt=0.002, forces=[pot], methods=[azp.ConstantVolume()])
        sim.operations.tuners.clear()  # the default sorter (period 200) would not fire in 20 steps anyway
        if period:
            sim.operations.tuners.append(azp.ParticleSorter(trigger_period=period, particles_per_block=64))
        sim.run(20)
        t = sim.state.tag.cpu().numpy().view(np.uint32)
        x = np.zeros((n, 3))
        x[t] = sim.state.pos[:, :3].cpu().numpy()
        out.append(x)
        if period:
            assert sim.operations.tuners[0].num_sorts == 4
    assert np.abs(out[0] - out[1]).max() < 1e-10


def _hilbert(cx, cy, cz, b):
    """Skilling's axes -> Hilbert index (restated in Python; checked on the CPU in test_host_cpu.py)."""
    X = [cx, cy, cz]
    M = 1 << (b - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for k in range(3):
            if X[k] & Q:
                X[0] ^= P
            else:
                t = (X[0] ^ X[k]) & P
                X[0] ^= t
                X[k] ^= t
        Q >>= 1
    X[1] ^= X[0]
    X[2] ^= X[1]
    t = 0
    Q = M
    while Q > 1:
        if X[2] & Q:
            t ^= Q - 1
        Q >>= 1
    X = [x ^ t for x in X]
    h = 0
    for bit in range(b - 1, -1, -1):
        h = (h << 3) | (((X[0] >> bit) & 1) << 2) | (((X[1] >> bit) & 1) << 1) | ((X[2] >> bit) & 1)
    return h


def test_hilbert_keys_are_the_hilbert_curve():
    """azp_sorter_keys with block = 0: dims (12, 9, 7) ask for a 16^3 grid stretched over the box; one particle at
    the centre of each of its cells: the keys are a bijection onto 0 .. 4095, consecutive keys are face neighbors
    (the curve never leaves the box) and equal the Python restatement."""
    import ctypes as C

    import torch

    from azplugins_amd import _lib

    dims = (12, 9, 7)
    side = 16
    L = np.array([6.0, 4.5, 3.5])
    cells = np.array([(x, y, z) for z in range(side) for y in range(side) for x in range(side)])
    xyz = (cells + 0.5) / side * L - 0.5 * L
    pos = torch.from_numpy(syn.pos4(xyz)).to("cuda:0")
    keys = torch.empty(len(cells), dtype=torch.int32, device="cuda:0")
    box = _lib.make_box(tuple(L))
    cdims = (C.c_uint32 * 3)(*dims)
    _lib.check(_lib.lib().azp_sorter_keys(len(cells), pos.data_ptr(), C.byref(box), cdims, 0, keys.data_ptr(), None), "azp_sorter_keys")
    torch.cuda.synchronize()
    got = keys.cpu().numpy()
    want = np.array([_hilbert(int(x), int(y), int(z), 4) for x, y, z in cells])
    assert np.array_equal(got, want) and sorted(got.tolist()) == list(range(side ** 3))
    order = cells[np.argsort(got)]
    assert (np.abs(np.diff(order, axis=0)).sum(axis=1) == 1).all()


def test_sorted_liquid_tiles_are_compact():
    """A jittered lattice in random memory order: after a Hilbert sort 256 consecutive particles form a
    compact blob wherever the run starts -- every tile can be staged, the plan is compiled straight from the
    cells, and the staged sets are smaller than with the row-major block order."""
    cfg = syn.config_plj_sc(28)
    n = cfg["xyz"].shape[0]
    perm = np.argsort(syn.u01(43, np.arange(n, dtype=np.uint64), 0))
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"][perm], cfg["L"]))
    nl = azp.nlist.Cell(buffer=0.4)
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=3.0)
    pot.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=0.001, forces=[pot])
    sim.operations.tuners.clear()
    azp.ParticleSorter(curve="hilbert").sort(sim)
    sim.run(0)
    info = pot.plan_info
    assert info["valid"] == 1 and info["from_cells"] == 1 and n % 256 != 0
    hilbert_mean = pot._plan.tile_stage().mean()
    azp.ParticleSorter(curve="blocks").sort(sim)
    sim.run(0)
    assert pot.plan_info["valid"] == 1
    assert hilbert_mean < pot._plan.tile_stage().mean()


def test_bond_members_host_and_device_mirrors():
    """State.bond_group (host array, HOOMD's snapshot layout) and State.bond_group_device() (int64 tensor) are two views
    of one table: writing either invalidates the other, the copy across happens on demand (the particle sorter re-indexes
    the bonds on the device and nothing travels inside a run)."""
    import torch

    from azplugins_amd.state import Snapshot, State

    xyz = np.arange(30, dtype=np.float64).reshape(10, 3) * 0.1
    bonds = np.array([[0, 1], [1, 2], [4, 9], [7, 3]], dtype=np.uint32)
    st = State(Snapshot.from_arrays(xyz, [10.0, 10.0, 10.0], bonds=bonds), "cuda:0")
    assert st.n_bonds == 4 and st.bond_group.dtype == np.uint32 and np.array_equal(st.bond_group, bonds)
    dev = st.bond_group_device()
    assert dev.dtype == torch.int64 and dev.shape == (4, 2) and np.array_equal(dev.cpu().numpy(), bonds.astype(np.int64))
    assert st.bond_group_device() is dev  # (cached)
    # re-indexed on the device (what the sorter does): the host mirror follows when asked
    inv = torch.tensor([9, 8, 7, 6, 5, 4, 3, 2, 1, 0], dtype=torch.int64, device="cuda:0")
    st.set_bond_group_device(inv[dev])
    assert st.n_bonds == 4
    assert np.array_equal(st.bond_group, (9 - bonds.astype(np.int64)).astype(np.uint32)) and st.bond_group.dtype == np.uint32
    # written on the host: the device copy is rebuilt
    st.bond_group = bonds[:2]
    assert st.n_bonds == 2 and np.array_equal(st.bond_group_device().cpu().numpy(), bonds[:2].astype(np.int64))
