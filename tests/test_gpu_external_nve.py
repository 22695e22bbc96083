"""Rows N4 / N2 of SURVEY 8f on the GPU: harmonic barriers (the reference's own
test_external.py cases, through the hoomd.azplugins.external-shaped API) and the
velocity-Verlet NVE kernels against the oracle."""

import json
import os

import numpy as np
import pytest

import azplugins_amd as azp
from azplugins_amd import synthetic as syn

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "reference_cases.json")) as _f:
    EXT = json.load(_f)["external"]


class CustomVariant:
    """src/pytest/test_external.py:17-33"""

    def __init__(self, z):
        self.z = float(z)

    def __call__(self, timestep):
        return self.z if timestep <= 1 else self.z - 1


def _integrator():
    ig = azp.Integrator(dt=0.0)
    ig.methods = [azp.ConstantVolume()]
    return ig


@pytest.mark.parametrize("cls", [azp.external.PlanarHarmonicBarrier, azp.external.SphericalHarmonicBarrier])
def test_create(cls):
    barrier = cls(location=3.0)
    barrier.params["A"].update(dict(k=10.0, offset=0.5))
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.two_particle_snapshot())
    ig = _integrator()
    sim.operations.integrator = ig
    ig.forces.append(barrier)
    assert barrier.params["A"] == dict(k=10.0, offset=0.5)
    sim.run(0)
    assert barrier.params["A"] == dict(k=10.0, offset=0.5)


@pytest.mark.parametrize("kind", ["spherical", "planar"])
def test_harmonic_barrier(kind):
    c = EXT[kind]
    snap = azp.Snapshot()
    snap.configuration.box = azp.Box.from_box([20, 20, 20, 0, 0, 0])
    snap.particles.N = 4
    snap.particles.types = ["A", "B"]
    snap.particles.position[:] = c["positions"]
    snap.particles.typeid[:] = EXT["typeid"]
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(snap)
    ig = _integrator()
    sim.operations.integrator = ig
    cls = azp.external.SphericalHarmonicBarrier if kind == "spherical" else azp.external.PlanarHarmonicBarrier
    barrier = cls(location=CustomVariant(z=5.0))
    barrier.params["A"] = dict(k=EXT["kA"], offset=EXT["offset_A"])
    barrier.params["B"] = dict(k=EXT["kB"], offset=EXT["offset_B"])
    ig.forces.append(barrier)
    sim.run(1)
    np.testing.assert_allclose(barrier.energies, c["run1"]["energies"], atol=1e-4)
    np.testing.assert_allclose(barrier.forces, c["run1"]["forces"], atol=1e-4)
    barrier.params["B"] = dict(k=0.0, offset=EXT["offset_B"])
    sim.run(2)
    np.testing.assert_allclose(barrier.energies, c["run2"]["energies"], atol=1e-4)
    np.testing.assert_allclose(barrier.forces, c["run2"]["forces"], atol=1e-4)


def test_barrier_invalid_location_raises():
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.two_particle_snapshot(L=20.0))
    ig = _integrator()
    sim.operations.integrator = ig
    barrier = azp.external.PlanarHarmonicBarrier(location=11.0)  # outside [-10, 10)
    barrier.params["A"] = dict(k=1.0, offset=0.0)
    ig.forces.append(barrier)
    with pytest.raises(azp.AzpError):
        sim.run(0)
    sph = azp.external.SphericalHarmonicBarrier(location=10.5)  # 2 R > L
    sph.params["A"] = dict(k=1.0, offset=0.0)
    ig.forces[:] = [sph]
    sim._attached.clear()
    with pytest.raises(azp.AzpError):
        sim.run(0)


def test_barrier_parity_large(oracle):
    """N = 32,768 random particles, 3 types, some outside the box (wrapped first)."""
    xyz, L, _ = syn.uniform_random(32768, 24.0, seed=3)
    xyz = xyz * 1.08  # ~8 % of the particles drift just outside: must be wrapped back
    typeid = np.arange(32768) % 3
    params = [[50.0, 0.1], [200.0, -0.4], [0.0, 0.0]]
    pos = syn.pos4(xyz, typeid)
    snap = azp.Snapshot.from_arrays(xyz, L, typeid=typeid, types=("A", "B", "C"))
    for kind, cls, loc in (("planar", azp.external.PlanarHarmonicBarrier, 3.0), ("spherical", azp.external.SphericalHarmonicBarrier, 9.0)):
        sim = azp.Simulation(device="cuda:0", seed=1)
        sim.create_state_from_snapshot(snap)
        b = cls(location=loc)
        for t, (k, off) in zip("ABC", params):
            b.params[t] = dict(k=k, offset=off)
        sim.operations.integrator = azp.Integrator(dt=0.0, forces=[b])
        sim.run(0)
        ref = oracle.barrier_forces(kind, pos, oracle.make_box(L), params, loc)
        got = np.c_[b.forces, b.energies]
        assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
        assert (ref[:, 3] > 0).sum() > 1000


def test_nve_steps_match_oracle(oracle):
    """Ten velocity-Verlet steps of a small PerturbedLJ fluid: GPU kernels (forces
    + integration) vs oracle forces + oracle integration, same lists each step."""
    cfg = syn.config_plj_sc(10)
    n = cfg["xyz"].shape[0]
    tag = np.arange(n, dtype=np.uint64)
    vel0 = np.stack([syn.normal(5, tag, c) for c in range(3)], axis=1) * 0.8
    vel0 -= vel0.mean(axis=0)
    snap = azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], velocity=vel0)
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(snap)
    nl = azp.nlist.Cell(buffer=0.4)
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=2.5, mode="shift")
    pot.params[("A", "A")] = cfg["params"]
    dt = 0.002
    sim.operations.integrator = azp.Integrator(dt=dt, forces=[pot], methods=[azp.ConstantVolume()])
    sim.run(10)
    # oracle trajectory
    box = oracle.make_box(cfg["L"])
    p = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    pos = syn.pos4(cfg["xyz"])
    vel = np.zeros((n, 4)); vel[:, :3] = vel0; vel[:, 3] = 1.0

    def forces(x):
        return oracle.pair_forces("PerturbedLennardJones", x, box, oracle.build_nlist(x, box, 2.9), p, 2.5, mode="shift")

    f = forces(pos)
    for _ in range(10):
        oracle.nve_step(True, pos, vel, f, box, dt)
        f = forces(pos)
        oracle.nve_step(False, pos, vel, f, box, dt)
    got_pos = sim.state.pos.cpu().numpy()
    got_vel = sim.state.vel.cpu().numpy()
    assert np.abs(got_pos[:, :3] - pos[:, :3]).max() < 1e-11
    assert np.abs(got_vel[:, :3] - vel[:, :3]).max() < 1e-10
    # energy conservation over the short run (sanity of the integrator itself)
    ke = 0.5 * (got_vel[:, :3] ** 2).sum()
    assert np.isfinite(ke) and nl.num_builds >= 1


def test_halo_pack_matches_index_select():
    """azp_halo_pack (send buffer of the ghost exchange) against torch.index_select."""
    import ctypes as C

    import torch

    from azplugins_amd import _lib

    g = torch.Generator().manual_seed(5)
    stream = torch.cuda.current_stream().cuda_stream
    for width in (4, 2, 6):
        src = torch.rand((5000, width), dtype=torch.float64, generator=g).cuda()
        idx = torch.randint(0, 5000, (3333,), generator=g).cuda()
        dst = torch.full((3333, width), float("nan"), dtype=torch.float64, device="cuda:0")
        _lib.check(_lib.lib().azp_halo_pack(idx.numel(), src.data_ptr(), idx.data_ptr(), width, dst.data_ptr(), stream), "pack")
        assert torch.equal(dst, src.index_select(0, idx))
    # a rank without peers packs nothing (null buffers are fine), odd row widths are rejected
    assert _lib.lib().azp_halo_pack(0, None, None, 4, None, stream) == 0
    assert _lib.lib().azp_halo_pack(10, src.data_ptr(), idx.data_ptr(), 3, dst.data_ptr(), stream) != 0


def test_rotational_nve_kernels_match_oracle(oracle):
    """azp_integrate_nve_rot_step_one / _two against the oracle's restatement on random
    orientations, angular momenta, torques and moments of inertia (some axes zero)."""
    import ctypes as C

    import torch

    from azplugins_amd import _lib

    rng = np.random.default_rng(11)
    n = 5000
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    p = rng.normal(size=(n, 4))
    I = rng.uniform(0.2, 1.5, size=(n, 3))
    I[::7, 2] = 0.0
    I[::11, 0] = 0.0
    tq = np.zeros((n, 4))
    tq[:, :3] = rng.normal(size=(n, 3))
    dq, dp, dI, dt_ = (torch.from_numpy(x).cuda() for x in (q, p, I, tq))
    a = _lib.NVERotArgs()
    a.d_orientation, a.d_angmom, a.d_inertia, a.d_net_torque = dq.data_ptr(), dp.data_ptr(), dI.data_ptr(), dt_.data_ptr()
    a.dt, a.N = 0.004, n
    stream = torch.cuda.current_stream().cuda_stream
    lib = _lib.lib()
    for _ in range(5):
        _lib.check(lib.azp_integrate_nve_rot_step_one(C.byref(a), stream), "rot step one")
        q, p = oracle.nve_rot_step(True, q, p, I, tq, 0.004)
        _lib.check(lib.azp_integrate_nve_rot_step_two(C.byref(a), stream), "rot step two")
        q, p = oracle.nve_rot_step(False, q, p, I, tq, 0.004)
    torch.cuda.synchronize()
    assert np.abs(dq.cpu().numpy() - q).max() < 1e-13
    assert np.abs(dp.cpu().numpy() - p).max() < 1e-12 * max(1.0, np.abs(p).max())
    assert _lib.lib().azp_integrate_nve_rot_step_one(None, stream) != 0


def test_two_patch_morse_nve_with_rotation_conserves_energy():
    """Translational + rotational NVE with TwoPatchMorse forces and torques
    (Integrator(integrate_rotational_dof=True); the reference's aniso test gives its
    particles a moment of inertia, src/pytest/test_pair_aniso.py:113-140): the total energy
    U + K_trans + K_rot is conserved while energy flows into the rotations. The quaternion
    update is PARITY UNPINNED against HOOMD (source absent); this is its physical check."""
    cfg = syn.config_tpm(10, 10, 10)
    n = cfg["xyz"].shape[0]
    tag = np.arange(n, dtype=np.uint64)
    vel = np.stack([syn.normal(77, tag, c) for c in range(3)], axis=1) * np.sqrt(0.05)
    vel -= vel.mean(axis=0)
    snap = azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], velocity=vel, orientation=cfg["orientation"],
                                    moment_inertia=np.tile([0.1, 0.1, 0.1], (n, 1)))
    energies = {}
    for dt in (0.002, 0.001):
        sim = azp.Simulation(device="cuda:0", seed=1)
        sim.create_state_from_snapshot(snap)
        nl = azp.nlist.Cell(buffer=0.4)
        # a soft well (M_r = 0.25) keeps the time step of a test affordable; repulsive core on.
        # mode "none" with a cutoff where U_Morse has decayed to -2e-5 M_d. (The reference's energy
        # shift subtracts U_Morse(r_cut) Omega_i Omega_j from the ENERGY only, src/
        # AnisoPairEvaluatorTwoPatchMorse.h:194-207 -- a term that varies with the orientations but
        # exerts no torque, so with mode "shift" U + K is not a constant of the motion: +21 over this
        # run. And with r_cut = 3.0, U(r_cut) = -1e-3, the 24-neighbor lattice shell at 2.94 drifting
        # out of the cutoff shows up as a time-step-independent +1.06: tools/rot_energy_probe.py.)
        tpm = azp.pair.TwoPatchMorse(nlist=nl, default_r_cut=4.0, mode="none")
        tpm.params[("A", "A")] = dict(M_d=1.0, M_r=0.25, r_eq=1.1, omega=5.0, alpha=0.4, repulsion=True)
        sim.operations.integrator = azp.Integrator(dt=dt, forces=[tpm], methods=[azp.ConstantVolume()], integrate_rotational_dof=True)
        sim.operations.tuners.clear()
        sim.run(0)

        def total():
            v = sim.state.vel[:n]
            k_t = 0.5 * float((v[:, 3] * (v[:, :3] ** 2).sum(dim=1)).sum().item())
            return tpm.energy + k_t + sim.rotational_kinetic_energy(), k_t, sim.rotational_kinetic_energy()

        e0, kt0, kr0 = total()
        assert kr0 == 0.0
        sim.run(int(round(1.0 / dt)))
        e1, kt1, kr1 = total()
        energies[dt] = (e0, e1, kt0, kr1)
        assert kr1 > 1e-3 * kt0  # the torques did spin the particles up
        q = sim.state.orientation.cpu().numpy()
        assert np.allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-12)
    for dt, (e0, e1, kt0, kr1) in energies.items():
        assert abs(e1 - e0) < 5e-3 * kt0, "dt=%g: energy drift %g vs K_trans(0) %g" % (dt, e1 - e0, kt0)


def test_nve_step_two_one_is_step_two_then_step_one():
    """azp_integrate_nve_step_two_one (what Simulation.run launches between two force evaluations) leaves
    exactly the bits that azp_integrate_nve_step_two followed by azp_integrate_nve_step_one leave: positions,
    velocities, images, particles wrapped through the periodic boundary included."""
    import ctypes as C

    import torch

    from azplugins_amd import _lib

    n = 5000
    tag = np.arange(n, dtype=np.uint64)
    L = 7.0
    pos = np.stack([(syn.u01(1, tag, c) - 0.5) * L for c in range(3)] + [np.zeros(n)], axis=1)
    vel = np.stack([syn.normal(2, tag, c) * 3.0 for c in range(3)] + [0.5 + syn.u01(3, tag, 0)], axis=1)  # w = mass
    frc = np.stack([syn.normal(4, tag, c) * 50.0 for c in range(4)], axis=1)
    results = []
    for fused in (False, True):
        t = dict(pos=torch.from_numpy(pos.copy()).to("cuda:0"), vel=torch.from_numpy(vel.copy()).to("cuda:0"),
                 frc=torch.from_numpy(frc.copy()).to("cuda:0"), image=torch.zeros((n, 3), dtype=torch.int32, device="cuda:0"))
        a = _lib.NVEArgs()
        a.d_pos, a.d_vel, a.d_net_force, a.d_image = t["pos"].data_ptr(), t["vel"].data_ptr(), t["frc"].data_ptr(), t["image"].data_ptr()
        a.box = _lib.make_box(L)
        a.dt = 0.05
        a.N = n
        lib = _lib.lib()
        if fused:
            _lib.check(lib.azp_integrate_nve_step_two_one(C.byref(a), None), "two_one")
        else:
            _lib.check(lib.azp_integrate_nve_step_two(C.byref(a), None), "two")
            _lib.check(lib.azp_integrate_nve_step_one(C.byref(a), None), "one")
        torch.cuda.synchronize()
        results.append({k: v.cpu().numpy().copy() for k, v in t.items()})
    for k in ("pos", "vel", "image"):
        assert np.array_equal(results[0][k], results[1][k]), k
    assert np.abs(results[0]["image"]).sum() > 0   # some particles did cross the boundary


def test_sum_forces_and_displacements():
    """azp_sum_forces (Integrator::computeNetForce in one pass) equals the element-wise sum; azp_nlist_displacements
    writes every particle's displacement as a single-precision UPPER bound next to the maximum and the rebuild flag."""
    import ctypes as C

    import torch

    from azplugins_amd import _lib

    lib = _lib.lib()
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda:0").manual_seed(3)
    n = 100_003
    fs = [torch.randn((n, 4), dtype=torch.float64, device="cuda:0", generator=g) for _ in range(3)]
    out = torch.full((n, 4), float("nan"), dtype=torch.float64, device="cuda:0")
    ptrs = (C.c_void_p * 3)(*[f.data_ptr() for f in fs])
    _lib.check(lib.azp_sum_forces(n, 3, ptrs, out.data_ptr(), stream), "azp_sum_forces")
    assert torch.equal(out, (fs[0] + fs[1]) + fs[2])
    # in place: the accumulator among the inputs
    ptrs2 = (C.c_void_p * 2)(out.data_ptr(), fs[0].data_ptr())
    ref = out + fs[0]
    _lib.check(lib.azp_sum_forces(n, 2, ptrs2, out.data_ptr(), stream), "azp_sum_forces")
    assert torch.equal(out, ref)

    L = np.array([30.0, 20.0, 25.0])
    rng = np.random.default_rng(9)
    x0 = (rng.random((n, 3)) - 0.5) * L
    d = rng.normal(size=(n, 3)) * 0.05
    d[17] = [0.31, 0.0, 0.0]           # the one particle beyond the limit
    x1 = syn.wrap(x0 + d, L)           # (some cross the periodic boundary: minimum image)
    p0 = torch.from_numpy(syn.pos4(x0)).to("cuda:0")
    p1 = torch.from_numpy(syn.pos4(x1)).to("cuda:0")
    box = _lib.make_box(L)
    row = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    disp = torch.full((n,), -1.0, dtype=torch.float32, device="cuda:0")
    _lib.check(lib.azp_nlist_displacements(n, p1.data_ptr(), p0.data_ptr(), C.byref(box), 0.3 ** 2, row.data_ptr(), row.data_ptr() + 8,
                                           disp.data_ptr(), stream), "azp_nlist_displacements")
    flag, bits = row.tolist()
    exact = np.linalg.norm(d, axis=1)
    got = disp.cpu().numpy().astype(np.float64)
    assert flag == 1
    assert abs(np.sqrt(np.array([bits], dtype=np.int64).view(np.float64)[0]) - exact.max()) < 1e-12
    assert np.all(got >= exact * (1 - 1e-15)) and np.all(got <= exact * (1 + 3e-7) + 1e-30)
