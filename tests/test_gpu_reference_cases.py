"""The reference's own known-answer tests (src/pytest/test_pair.py,
test_pair_aniso.py, test_bond.py), re-run against the MI355X path through the
Python mirror of hoomd.azplugins.pair / bond. Same set-up, same observables,
same tolerance (4 decimals) as the reference."""

import json
import os

import numpy as np
import pytest

import azplugins_amd as azp

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "reference_cases.json")) as _f:
    CASES = json.load(_f)


def _ids(cases):
    return ["%s@%s" % (c["potential"], c["src"].split(":")[-1]) for c in cases]


@pytest.mark.parametrize("case", CASES["pair"], ids=_ids(CASES["pair"]))
def test_energy_and_force(case):
    # make 2 particle test configuration (src/pytest/test_pair.py:316-322)
    r_cut = case["r_cut"]
    r_buff = 0.4
    L_domain_min = 2 * (r_cut + r_buff)
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.two_particle_snapshot(d=case["distance"], L=2.1 * L_domain_min))

    integrator = azp.Integrator(dt=0.001)
    integrator.methods = [azp.ConstantVolume()]

    cls = getattr(azp.pair, case["potential"])
    extra_args = {}
    if cls is azp.pair.DPDGeneralWeight:
        extra_args["kT"] = 0.0
    else:
        extra_args["mode"] = "shift" if case["shift"] else "none"
    potential = cls(nlist=azp.nlist.Cell(buffer=r_buff), default_r_cut=r_cut, **extra_args)
    potential.params[("A", "A")] = case["params"]
    integrator.forces = [potential]

    sim.operations.integrator = integrator
    sim.run(0)

    # parameters are still correct after attach runs (round trip through the C structs)
    assert potential.params[("A", "A")] == case["params"]

    e = case["energy"]
    np.testing.assert_array_almost_equal(potential.energies, [0.5 * e, 0.5 * e], decimal=4)
    f = case["force"]
    np.testing.assert_array_almost_equal(potential.forces, [[-f, 0, 0], [f, 0, 0]], decimal=4)


@pytest.mark.parametrize("case", CASES["aniso"], ids=_ids(CASES["aniso"]))
def test_energy_force_and_torque(case):
    snap = azp.two_particle_snapshot()
    snap.particles.position[:] = CASES["aniso_setup"]["positions"]
    snap.particles.orientation[:] = CASES["aniso_setup"]["orientations"]
    snap.particles.moment_inertia[:] = [0.1, 0.1, 0.1]
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(snap)

    integrator = azp.Integrator(dt=0.001)
    integrator.methods = [azp.ConstantVolume()]
    potential = azp.pair.TwoPatchMorse(nlist=azp.nlist.Cell(buffer=0.4), default_r_cut=case["r_cut"],
                                       mode="shift" if case["shift"] else "none")
    potential.params[("A", "A")] = case["params"]
    integrator.forces = [potential]
    sim.operations.integrator = integrator
    sim.run(0)

    ref_values = list(case["params"].values())
    test_values = [potential.params[("A", "A")][k] for k in case["params"]]
    assert np.allclose(test_values, ref_values)

    e = case["energy"]
    np.testing.assert_array_almost_equal(potential.energies, [0.5 * e, 0.5 * e], decimal=4)
    if case["force"] is not None:
        f = np.array(case["force"])
        np.testing.assert_array_almost_equal(potential.forces, [-f, f], decimal=4)
    if case["torque"] is not None:
        T = np.array(case["torque"])
        np.testing.assert_array_almost_equal(potential.torques, [T, T], decimal=4)


@pytest.mark.parametrize("case", CASES["bond"], ids=_ids(CASES["bond"]))
def test_bond_energy_and_force(case):
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.bonded_two_particle_snapshot(d=case["distance"]))
    integrator = azp.Integrator(dt=0.001)
    integrator.methods = [azp.ConstantVolume()]
    potential = getattr(azp.bond, case["potential"])()
    potential.params["A-A"] = case["params"]
    integrator.forces = [potential]
    sim.operations.integrator = integrator
    sim.run(0)

    assert potential.params["A-A"] == case["params"]
    e = case["energy"]
    np.testing.assert_array_almost_equal(potential.energies, [0.5 * e, 0.5 * e], decimal=4)
    f = case["force"]
    np.testing.assert_array_almost_equal(potential.forces, [[-f, 0, 0], [f, 0, 0]], decimal=4)


def test_bond_invalid_parameters_raise():
    """Evaluator returning false => HOOMD's 'bond out of bounds' error
    (src/PotentialBondGPUKernel.cu.inc:29, src/BondEvaluatorDoubleWell.h:101-102)."""
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.bonded_two_particle_snapshot(d=1.0))
    integrator = azp.Integrator(dt=0.001)
    potential = azp.bond.DoubleWell()
    potential.params["A-A"] = dict(r_0=1.0, r_1=1.0, U_1=1.0, U_tilt=0.0)
    integrator.forces = [potential]
    sim.operations.integrator = integrator
    with pytest.raises(azp.AzpError):
        sim.run(0)


def test_bond_invalid_parameters_raise_inside_a_run():
    """The same error when the parameters go bad in the middle of a run: inside Simulation.run the bond evaluators' flag is
    examined when the NEXT step is queued and once more after the loop (no host round trip behind every launch), so the
    run still ends in the error; and a force with sane parameters again works afterwards."""
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.bonded_two_particle_snapshot(d=1.0))
    integrator = azp.Integrator(dt=0.001, methods=[azp.ConstantVolume()])
    potential = azp.bond.DoubleWell()
    potential.params["A-A"] = dict(r_0=1.0, r_1=1.5, U_1=1.0, U_tilt=0.5)
    integrator.forces = [potential]
    sim.operations.integrator = integrator
    sim.run(3)
    potential.params["A-A"] = dict(r_0=1.0, r_1=1.0, U_1=1.0, U_tilt=0.0)
    with pytest.raises(azp.AzpError):
        sim.run(5)
    potential.params["A-A"] = dict(r_0=1.0, r_1=1.5, U_1=1.0, U_tilt=0.5)
    sim.run(2)
    assert np.all(np.isfinite(potential.forces))


def test_dpd_temperature():
    """src/pytest/test_pair_dpd.py:13-46 restated: N=1000 on a 10^3 lattice
    (a=0.6), thermalised at kT=1.5, DPD with A=0 (drag + random only), NVE,
    dt=0.01; the mean kinetic temperature over 100 steps stays at 1.5 +- 10 %."""
    sim = azp.Simulation(device="cuda:0", seed=42)
    sim.create_state_from_snapshot(azp.lattice_snapshot(n=10, a=0.6))
    sim.thermalize_particle_momenta(kT=1.5)
    integrator = azp.Integrator(dt=0.01)
    sim.operations.integrator = integrator
    cell = azp.nlist.Cell(buffer=0.4)
    dpd = azp.pair.DPDGeneralWeight(nlist=cell, kT=1.5, default_r_cut=1.0)
    dpd.params[("A", "A")] = dict(A=0.0, gamma=4.5, s=0.5)
    integrator.forces.append(dpd)
    integrator.methods.append(azp.ConstantVolume())
    sim.run(10)
    num_samples = 100
    kT = np.zeros(num_samples)
    for sample in range(num_samples):
        kT[sample] = sim.kinetic_temperature()
        sim.run(1)
    assert np.mean(kT) == pytest.approx(1.5, 0.1)
