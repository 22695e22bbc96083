"""BASELINE.json configs[2..4] at their FULL sizes on one MI355X (C4 and C5 are multi-GPU
configurations in BASELINE.json; they fit one GPU, and their decomposed form is covered by
tests/test_domain.py and tests/test_gpu_domain.py): the product path through the
hoomd.azplugins-shaped API against

* the committed oracle fixtures (tests/golden/c{3,4,5}_sample.npz, written by
  tests/golden/make_golden.py full): every (N / 1024)-th particle's force (and torque) to
  1e-10 of the largest component, and the sums over all particles;
* size-independent properties: total force = 0 (third law through full lists; for DPD the
  pairwise noise is antisymmetric because both ends of a pair draw the same number), the
  tile-staged kernel agrees with the generic one, a second launch is bit-identical."""

import os

import numpy as np
import pytest

from azplugins_amd import synthetic as syn

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-10


def _check_against_fixture(name, *arrays):
    g = np.load(os.path.join(GOLDEN, name))
    idx = g["sample_index"]
    for k, a in enumerate(arrays):
        ref = g["sample_%d" % k]
        scale = max(np.abs(ref).max(), 1e-300)
        assert np.abs(a[idx] - ref).max() <= TOL * scale, "%s array %d: sample differs by %g (scale %g)" % (
            name, k, np.abs(a[idx] - ref).max(), scale)
        # sums over all N particles: rounding grows like sqrt(N) eps sum|x|
        tol = 1e-11 * g["abssum_%d" % k] + 1e-300
        assert np.all(np.abs(a.sum(axis=0) - g["sum_%d" % k]) <= tol), "%s array %d: sums differ" % (name, k)
        assert np.all(np.abs(np.abs(a).sum(axis=0) - g["abssum_%d" % k]) <= tol)
    return g


def _sim(cfg, **snap_kw):
    import azplugins_amd as azp

    sim = azp.Simulation(device="cuda:0", seed=cfg.get("seed", 1))
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], **snap_kw))
    return azp, sim


def _total_force_is_zero(f, what):
    scale = np.abs(f[:, :3]).max()
    assert np.abs(f[:, :3].sum(axis=0)).max() <= 1e-9 * scale * np.sqrt(f.shape[0]) / 1e3, what


def test_c3_chains_full_size():
    """32,768 chains of 32 beads, N = 1,048,576: PerturbedLJ (bonded pairs excluded from
    the list, HOOMD's default) + DoubleWell bonds."""
    cfg = syn.config_chains()
    azp, sim = _sim(cfg, bonds=cfg["bonds"])
    assert cfg["xyz"].shape[0] == 1048576 and len(cfg["bonds"]) == 1015808
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    plj = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=cfg["r_cut"], mode="shift")
    plj.params[("A", "A")] = cfg["params"]
    dw = azp.bond.DoubleWell()
    dw.params["A-A"] = cfg["bond_params"]
    sim.operations.integrator = azp.Integrator(dt=0.001, forces=[plj, dw])
    sim.run(0)
    assert plj.plan_info["valid"] == 1
    f_pair = np.c_[plj.forces, plj.energies]
    f_bond = np.c_[dw.forces, dw.energies]
    g = _check_against_fixture("c3_sample.npz", f_pair, f_bond)
    # rows compiled straight from the cells hold the exact list plus a hair (single-precision
    # acceptance test with a 1e-5 margin on r_list^2; the force kernel's FP64 cutoff test ignores them)
    extra = nl.n_pairs / f_pair.shape[0] - float(g["mean_neighbors"])
    assert plj.plan_info["from_cells"] == 1 and -1e-9 < extra < 1e-4 * float(g["mean_neighbors"])
    _total_force_is_zero(f_pair, "pair")
    _total_force_is_zero(f_bond, "bond")
    # generic kernel == tile kernel, and a relaunch is bit-identical
    first = plj.force_tensor.clone()
    plj.compute(0)
    assert bool((plj.force_tensor == first).all())
    plj.use_plan = False
    plj.compute(0)
    scale = float(first.abs().max())
    assert float((plj.force_tensor - first).abs().max()) <= 1e-11 * scale


def test_c4_dpd_full_size():
    """DPD thermostat, N = 2,097,152, rho = 3, r_cut = 1 (seed 7, timestep 0)."""
    cfg = syn.config_dpd()
    azp, sim = _sim(cfg, velocity=cfg["vel"], tag=cfg["tag"])
    assert cfg["xyz"].shape[0] == 2097152
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    dpd = azp.pair.DPDGeneralWeight(nlist=nl, kT=cfg["kT"], default_r_cut=cfg["r_cut"])
    dpd.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=cfg["dt"], forces=[dpd])
    sim.run(0)
    assert dpd.plan_info["valid"] == 1  # the tile-staged kernel ran
    f = np.c_[dpd.forces, dpd.energies]
    _check_against_fixture("c4_sample.npz", f)
    _total_force_is_zero(f, "dpd")
    first = dpd.force_tensor.clone()
    dpd.compute(0)
    assert bool((dpd.force_tensor == first).all())
    dpd.use_plan = False
    dpd.compute(0)
    assert float((dpd.force_tensor - first).abs().max()) <= 1e-11 * float(first.abs().max())
    # another timestep draws other noise; the conservative energy stays
    dpd.compute(1)
    assert not bool((dpd.force_tensor[:, :3] == first[:, :3]).all())
    assert float((dpd.force_tensor[:, 3] - first[:, 3]).abs().max()) <= 1e-12 * float(first[:, 3].abs().max())


def test_c5_two_patch_morse_full_size():
    """TwoPatchMorse patchy colloids, N = 524,288 with orientations (mode shift)."""
    cfg = syn.config_tpm()
    azp, sim = _sim(cfg, orientation=cfg["orientation"])
    assert cfg["xyz"].shape[0] == 524288
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    tpm = azp.pair.TwoPatchMorse(nlist=nl, default_r_cut=cfg["r_cut"], mode="shift")
    tpm.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=0.005, forces=[tpm])
    sim.run(0)
    assert tpm.plan_info["valid"] == 1
    f = np.c_[tpm.forces, tpm.energies]
    t = np.c_[tpm.torques, np.zeros(f.shape[0])]
    _check_against_fixture("c5_sample.npz", f, t)
    _total_force_is_zero(f, "tpm")
    first_f, first_t = tpm.force_tensor.clone(), tpm.torque_tensor.clone()
    tpm.compute(0)
    assert bool((tpm.force_tensor == first_f).all()) and bool((tpm.torque_tensor == first_t).all())
    tpm.use_plan = False
    tpm.compute(0)
    assert float((tpm.force_tensor - first_f).abs().max()) <= 1e-11 * float(first_f.abs().max())
    assert float((tpm.torque_tensor - first_t).abs().max()) <= 1e-11 * float(first_t.abs().max())


def test_north_star_liquid_after_sort_and_rebuilds(oracle):
    """N = 1,048,576 as an MD run leaves it: the lattice melted for 60 NVE steps at kT = 1
    (six list rebuilds on HOOMD's criterion), the particles re-indexed by the sorter (tiles now
    straddle the sorter's blocks), 12 more steps. The forces of the product path -- plan compiled
    from the cell list at every rebuild, rows stopped at the displacement bound -- against the
    oracle on its own list at the same positions, all particles."""
    import azplugins_amd as azp

    cfg = syn.config_north_star(64)
    azp_, sim = _sim(cfg)
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=cfg["r_cut"], mode="shift")
    pot.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot], methods=[azp.ConstantVolume()])
    sim.operations.tuners.clear()
    sim.run(0)
    sim.thermalize_particle_momenta(1.0, seed=11)
    sim.run(60)
    azp.ParticleSorter().sort(sim)
    sim.run(12)
    assert nl.num_builds >= 7 and pot.plan_info["valid"] == 1 and pot.plan_info["from_cells"] == 1
    assert nl.displacement_bound(sim.state) is not None
    n = sim.state.N
    pos = sim.state.pos[:n].cpu().numpy()
    box = oracle.make_box(cfg["L"])
    onl = oracle.build_nlist(pos, box, cfg["r_cut"])
    params = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, onl, params, cfg["r_cut"], 0.0, "shift", nthreads=8)
    f = np.c_[pot.forces, pot.energies]
    scale = np.abs(f_ref).max()
    assert np.abs(f - f_ref).max() <= 1e-10 * scale
    _total_force_is_zero(f, "pair")
