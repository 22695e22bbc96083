"""Parity of the gfx950 kernels (called through the C ABI) with the CPU oracle on
identical seeded snapshots.

Tolerances (FP64 path): forces / energies / virials / torques agree with the
oracle to TOL = 1e-10 of the largest component of the array (the two differ
only by FMA contraction, the Newton-refined reciprocal and summation order);
the north-star bar of 1e-5 relative is asserted per particle as well.
"""

import numpy as np
import pytest

import helpers as H
from azplugins_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TOL = 1e-10
NORTH_STAR_TOL = 1e-5


def assert_close(got, ref, tol=TOL, what=""):
    got = np.asarray(got)
    ref = np.asarray(ref)
    assert got.shape == ref.shape, what
    assert np.all(np.isfinite(got)), "%s: non-finite output" % what
    scale = np.abs(ref).max()
    err = np.abs(got - ref).max()
    assert err <= tol * (scale if scale > 0 else 1.0), "%s: max abs err %g vs scale %g" % (what, err, scale)


def assert_per_particle(got, ref):
    """north_star: forces within 1e-5 relative of the CPU reference, per
    particle (floor: 1e-3 of the largest force so exact zeros do not divide)."""
    n_ref = np.linalg.norm(ref[:, :3], axis=1)
    n_err = np.linalg.norm(got[:, :3] - ref[:, :3], axis=1)
    floor = 1e-3 * n_ref.max()
    assert np.all(n_err <= NORTH_STAR_TOL * np.maximum(n_ref, floor))


PAIR_PARAMS = {
    "PerturbedLennardJones": lambda i, j: dict(epsilon=1.0 + 0.15 * (i + j), sigma=1.0 - 0.03 * (i + j),
                                               attraction_scale_factor=0.5 - 0.1 * min(i, j)),
    "Hertz": lambda i, j: dict(epsilon=2.0 + i + j),
    "ExpandedYukawa": lambda i, j: dict(epsilon=1.0 + 0.5 * (i + j), kappa=1.2, delta=0.1 * (i + j)),
    "Colloid": None,  # below
    "DPDConservative": lambda i, j: dict(A=25.0 - 3 * (i + j), gamma=4.5, s=0.5),
}


def _params_table(oracle, name, T):
    if name == "Colloid":
        # type 0 = solvent (a=0), types >= 1 = colloids of radius 0.3, 0.4: covers the
        # solvent-solvent, colloid-solvent and colloid-colloid branches
        radius = [0.0, 0.3, 0.4]

        def fn(i, j):
            return dict(A=40.0 + 5 * (i + j), a_1=radius[i], a_2=radius[j], sigma=0.5)
    else:
        fn = PAIR_PARAMS[name]
    tab = H.sym_table(T, fn)
    # Colloid needs (a_1, a_2) stored symmetrically, as HOOMD's setParams does
    return np.array([oracle.pack_pair_params(name, tab[min(i, j)][max(i, j)]) for i in range(T) for j in range(T)])


def _config(oracle, T, name):
    # lattice spacing 1.1 with jitter keeps r_ij >~ 0.85: all potentials finite
    a = 1.1 if name != "Colloid" else 1.6
    pos, L, typeid = H.lattice_config(10, a, 0.1 * a, seed=21, ntypes=T)
    return pos, L


@pytest.mark.parametrize("name", ["PerturbedLennardJones", "Hertz", "ExpandedYukawa", "Colloid", "DPDConservative"])
@pytest.mark.parametrize("T", [1, 3])
@pytest.mark.parametrize("mode", ["none", "shift", "xplor"])
def test_pair_parity(oracle, name, T, mode):
    pos, L = _config(oracle, T, name)
    box = oracle.make_box(L)
    r_cut = np.full((T, T), 2.5 if name != "Colloid" else 3.2)
    if T > 1:
        r_cut[0, 1] = r_cut[1, 0] = r_cut[0, 0] - 0.3
    r_on = 0.8 * r_cut
    if T > 1 and mode == "xplor":
        r_on[2, 2] = r_cut[2, 2] + 0.1  # r_on > r_cut: xplor degenerates to shift for this pair
    r_buff = 0.3
    params = _params_table(oracle, name, T)
    nl_full = oracle.build_nlist(pos, box, r_cut + r_buff, ntypes=T)
    nl_half = oracle.build_nlist(pos, box, r_cut + r_buff, ntypes=T, half=True)
    # reference semantics: HOOMD CPU loop = half list + third-law scatter
    f_ref, v_ref = oracle.pair_forces(name, pos, box, nl_half, params, r_cut, r_on, mode, ntypes=T, half=True, virial=True)
    f_gpu, v_gpu = H.gpu_pair_forces(name, pos, (L,), nl_full, params, r_cut, r_on, mode, ntypes=T, virial=True)
    assert_close(f_gpu[:, :3], f_ref[:, :3], what="force")
    assert_close(f_gpu[:, 3], f_ref[:, 3], what="energy")
    assert_close(v_gpu, v_ref, what="virial")
    assert_per_particle(f_gpu, f_ref)
    # without the virial the force/energy must be bit-identical to the virial build
    f2 = H.gpu_pair_forces(name, pos, (L,), nl_full, params, r_cut, r_on, mode, ntypes=T, virial=False)
    assert np.array_equal(f2, f_gpu)


@pytest.mark.parametrize("tpp", [1, 2, 4, 8, 16, 32])
@pytest.mark.parametrize("block_size", [64, 256])
def test_pair_launch_shapes(oracle, tpp, block_size):
    """Every threads-per-particle / block-size variant gives the oracle's answer;
    N is not a multiple of anything convenient."""
    cfg = syn.config_plj_sc(9)
    pos = syn.pos4(cfg["xyz"][:-5])
    L = cfg["L"]
    box = oracle.make_box(L)
    params = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    nl = oracle.build_nlist(pos, box, 3.0 + 0.4)
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, nl, params, 3.0, mode="shift")
    f_gpu = H.gpu_pair_forces("PerturbedLennardJones", pos, (L,), nl, params, 3.0, mode="shift", tpp=tpp,
                              block_size=block_size)
    assert_close(f_gpu, f_ref)


def test_interior_skip_is_bit_identical(oracle):
    """The r_list_max hint (interior waves skip the minimum image) must not
    change a single bit."""
    cfg = syn.config_plj_sc(14)
    pos = syn.pos4(cfg["xyz"])
    L = cfg["L"]
    box = oracle.make_box(L)
    params = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    nl = oracle.build_nlist(pos, box, 3.4)
    f0 = H.gpu_pair_forces("PerturbedLennardJones", pos, (L,), nl, params, 3.0, r_list_max=0.0)
    f1 = H.gpu_pair_forces("PerturbedLennardJones", pos, (L,), nl, params, 3.0, r_list_max=3.4)
    assert np.array_equal(f0, f1)
    assert_close(f1, oracle.pair_forces("PerturbedLennardJones", pos, box, nl, params, 3.0))


def test_ghosts_and_nonperiodic(oracle):
    """Forces only for the N local particles; neighbors may be ghosts (index >=
    N); a non-periodic axis is not wrapped."""
    cfg = syn.config_plj_sc(10)
    xyz = cfg["xyz"]
    L = cfg["L"]
    order = np.argsort(xyz[:, 0] > 0.0, kind="stable")  # x <= 0 first: "local", rest "ghost"
    pos = syn.pos4(xyz[order])
    N = int((xyz[:, 0] <= 0.0).sum())
    box_o = oracle.make_box(L, periodic=(0, 1, 1))
    params = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    nl = oracle.build_nlist(pos, box_o, 2.9, N=N)
    assert nl[2].max() >= N
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box_o, nl, params, 2.5, N=N)
    f_gpu = H.gpu_pair_forces("PerturbedLennardJones", pos, (L, (0, 0, 0), (0, 1, 1)), nl, params, 2.5, N=N)
    assert f_gpu.shape == (N, 4)
    assert_close(f_gpu, f_ref)


def test_triclinic_box(oracle):
    cfg = syn.config_plj_sc(10)
    pos = syn.pos4(cfg["xyz"])
    L = cfg["L"]
    tilt = (0.2, -0.1, 0.15)
    box_o = oracle.make_box(L, tilt=tilt)
    params = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    # neighbor candidates from the orthorhombic list with a generous radius; the
    # kernels re-evaluate the true triclinic minimum image
    nl = oracle.build_nlist(pos, oracle.make_box(L), 4.5)
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box_o, nl, params, 2.0, mode="shift")
    f_gpu = H.gpu_pair_forces("PerturbedLennardJones", pos, (L, tilt), nl, params, 2.0, mode="shift")
    assert_close(f_gpu, f_ref)


def test_empty_rows_and_empty_system(oracle):
    """Particles with no neighbors get zeros (not stale memory); N = 0 is a no-op."""
    xyz = np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [5.0, 5.0, 5.0]])
    pos = syn.pos4(xyz)
    box = oracle.make_box(20.0)
    params = oracle.pack_pair_params("Hertz", dict(epsilon=1.0))
    nl = oracle.build_nlist(pos, box, 1.9)
    assert nl[0].tolist() == [1, 1, 0]
    f = H.gpu_pair_forces("Hertz", pos, (20.0,), nl, params, 1.5, virial=True)
    assert_close(f[0], oracle.pair_forces("Hertz", pos, box, nl, params, 1.5))
    assert not f[0][2].any() and not f[1][:, 2].any()
    import ctypes as C

    from azplugins_amd import _lib

    a = _lib.PairArgs()
    assert _lib.lib().azp_pair_forces_hertz(C.byref(a), 1, None) == 0  # N = 0
    assert _lib.lib().azp_pair_forces_hertz(None, None, None) == -1


def test_hertz_c1_golden(oracle):
    """BASELINE.json configs[0]: Hertz, N=4,096 random soft spheres, r_cut=1.0,
    against the oracle and the committed golden fixture."""
    import os

    cfg = syn.config_c1()
    pos = syn.pos4(cfg["xyz"])
    box = oracle.make_box(cfg["L"])
    params = oracle.pack_pair_params("Hertz", cfg["params"])
    nl = oracle.build_nlist(pos, box, cfg["r_cut"] + cfg["r_buff"])
    f_ref = oracle.pair_forces("Hertz", pos, box, nl, params, cfg["r_cut"], half=False)
    f_gpu = H.gpu_pair_forces("Hertz", pos, (cfg["L"],), nl, params, cfg["r_cut"])
    assert_close(f_gpu, f_ref)
    assert_per_particle(f_gpu, f_ref)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "c1_hertz_forces.npz"))
    assert_close(f_gpu, gold["force"], tol=1e-12)
    assert nl[0].mean() == pytest.approx(float(gold["mean_neighbors"]))


def test_dpd_thermostat_parity(oracle):
    """Drag + random + conservative forces with the same Philox stream."""
    cfg = syn.config_dpd(4096)
    for T in (1, 2):
        n = cfg["xyz"].shape[0]
        typeid = (np.arange(n) % T) if T > 1 else None
        pos = syn.pos4(cfg["xyz"], typeid)
        vel = np.zeros((n, 4))
        vel[:, :3] = cfg["vel"]
        vel[:, 3] = 1.0
        box = oracle.make_box(cfg["L"])
        tab = H.sym_table(T, lambda i, j: dict(A=25.0 - 5 * (i + j), gamma=4.5 + i + j, s=[0.5, 1.0, 2.0][i + j]))
        params = np.array([oracle.pack_pair_params("DPDGeneralWeight", tab[i][j]) for i in range(T) for j in range(T)])
        nl_f = oracle.build_nlist(pos, box, 1.4, ntypes=T)
        nl_h = oracle.build_nlist(pos, box, 1.4, ntypes=T, half=True)
        kw = dict(kT=1.0, dt=0.01, seed=7, timestep=123456789, ntypes=T)
        f_ref, v_ref = oracle.dpd_forces(pos, vel, cfg["tag"], box, nl_h, params, 1.0, half=True, virial=True, **kw)
        f_gpu, v_gpu = H.gpu_dpd_forces(pos, vel, cfg["tag"], (cfg["L"],), nl_f, params, 1.0, virial=True, **kw)
        assert_close(f_gpu[:, :3], f_ref[:, :3], what="dpd force")
        assert_close(f_gpu[:, 3], f_ref[:, 3], what="dpd energy")
        assert_close(v_gpu, v_ref, what="dpd virial")
        # pairwise noise is antisymmetric: total momentum is conserved
        assert np.abs(f_gpu[:, :3].sum(axis=0)).max() < 1e-9 * np.abs(f_gpu[:, :3]).max()
        # a different timestep or seed draws different noise
        kw2 = dict(kw, timestep=123456790)
        f_other = H.gpu_dpd_forces(pos, vel, cfg["tag"], (cfg["L"],), nl_f, params, 1.0, **kw2)
        assert not np.allclose(f_other[:, :3], f_gpu[:, :3])
        assert np.array_equal(f_other[:, 3], f_gpu[:, 3])  # energy is conservative only


def test_dpd_kT_zero_is_deterministic_part(oracle):
    cfg = syn.config_dpd(2048)
    n = cfg["xyz"].shape[0]
    pos = syn.pos4(cfg["xyz"])
    vel = np.zeros((n, 4))
    vel[:, 3] = 1.0
    box = oracle.make_box(cfg["L"])
    p = oracle.pack_pair_params("DPDGeneralWeight", cfg["params"])
    nl = oracle.build_nlist(pos, box, 1.4)
    f_thermo = H.gpu_dpd_forces(pos, vel, cfg["tag"], (cfg["L"],), nl, p, 1.0, kT=0.0, dt=0.01, seed=1, timestep=5)
    f_cons = H.gpu_pair_forces("DPDConservative", pos, (cfg["L"],), nl, p, 1.0)
    assert_close(f_thermo, f_cons, tol=1e-13)


def test_aniso_parity(oracle):
    cfg = syn.config_tpm(8, 8, 12)
    n = cfg["xyz"].shape[0]
    for T, mode in ((1, "none"), (1, "shift"), (2, "shift")):
        typeid = (np.arange(n) % T) if T > 1 else None
        pos = syn.pos4(cfg["xyz"], typeid)
        box = oracle.make_box(cfg["L"])
        tab = H.sym_table(T, lambda i, j: dict(cfg["params"], M_d=1.8341 + 0.2 * (i + j), repulsion=bool((i + j) % 2)))
        params = np.array([oracle.pack_pair_params("TwoPatchMorse", tab[i][j]) for i in range(T) for j in range(T)])
        nl_f = oracle.build_nlist(pos, box, 2.0, ntypes=T)
        nl_h = oracle.build_nlist(pos, box, 2.0, ntypes=T, half=True)
        f_ref, t_ref, v_ref = oracle.aniso_forces_tpm(pos, cfg["orientation"], box, nl_h, params, 1.6, mode, ntypes=T,
                                                      half=True, virial=True)
        f_gpu, t_gpu, v_gpu = H.gpu_aniso_forces(pos, cfg["orientation"], (cfg["L"],), nl_f, params, 1.6, mode, ntypes=T,
                                                 virial=True)
        assert_close(f_gpu[:, :3], f_ref[:, :3], what="aniso force")
        assert_close(f_gpu[:, 3], f_ref[:, 3], what="aniso energy")
        assert_close(t_gpu[:, :3], t_ref[:, :3], what="aniso torque")
        assert_close(v_gpu, v_ref, what="aniso virial")
        assert not t_gpu[:, 3].any()
        assert np.abs(f_ref[:, :3]).max() > 1e-3 and np.abs(t_ref[:, :3]).max() > 1e-3


@pytest.mark.parametrize("name", ["DoubleWell", "Quartic"])
def test_bond_parity(oracle, name):
    cfg = syn.config_chains(32, 6, 6, 16)
    pos = syn.pos4(cfg["xyz"])
    box = oracle.make_box(cfg["L"])
    bonds = cfg["bonds"]
    btype = (np.arange(bonds.shape[0]) % 2).astype(np.uint32)
    if name == "DoubleWell":
        ps = [dict(r_0=1.0, r_1=1.5, U_1=1.0, U_tilt=0.5), dict(r_0=0.9, r_1=1.3, U_1=2.0, U_tilt=0.0)]
    else:
        ps = [dict(k=1434.3, r_0=1.5, b_1=-0.7589, b_2=0.0, U_0=67.2234, sigma=1.0, epsilon=1.0, delta=0.0),
              dict(k=1000.0, r_0=1.6, b_1=-0.5, b_2=0.1, U_0=50.0, sigma=0.9, epsilon=1.2, delta=0.15)]
    params = np.array([oracle.pack_bond_params(name, p) for p in ps])
    f_ref, bad, v_ref = oracle.bond_forces(name, pos, box, bonds, btype, params, virial=True)
    assert bad == 0
    f_gpu, flag, v_gpu = H.gpu_bond_forces(name, pos, (cfg["L"],), bonds, btype, params, virial=True)
    assert flag == 0
    assert_close(f_gpu[:, :3], f_ref[:, :3], what="bond force")
    assert_close(f_gpu[:, 3], f_ref[:, 3], what="bond energy")
    assert_close(v_gpu, v_ref, what="bond virial")
    # invalid parameters raise the flag and contribute nothing
    bad_p = params.copy()
    if name == "DoubleWell":
        bad_p[1, 1] = 0.0  # r_diff = 0
    else:
        bad_p[1, 1] = 0.0  # r_0 = 0
    f_bad, flag = H.gpu_bond_forces(name, pos, (cfg["L"],), bonds, btype, bad_p)
    assert flag == 1
    f_ref_bad, nbad = oracle.bond_forces(name, pos, box, bonds, btype, bad_p)
    assert nbad == (btype == 1).sum()
    assert_close(f_bad, f_ref_bad)


def test_plj_c2_full_size(oracle):
    """BASELINE.json configs[1] at full size: PerturbedLJ, N=262,144, rho*=0.8,
    r_cut=3.0 vs the oracle (OpenMP full-list loop, same arithmetic per pair)
    plus size-independent properties."""
    cfg = syn.config_plj_sc(64)
    pos = syn.pos4(cfg["xyz"])
    box = oracle.make_box(cfg["L"])
    params = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    nl = oracle.build_nlist(pos, box, 3.4)
    assert nl[0].mean() == pytest.approx(131.7, abs=2.0)
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, nl, params, 3.0, mode="shift", nthreads=16)
    f_gpu = H.gpu_pair_forces("PerturbedLennardJones", pos, (cfg["L"],), nl, params, 3.0, mode="shift", r_list_max=3.4)
    assert_close(f_gpu, f_ref)
    assert_per_particle(f_gpu, f_ref)
    # Newton's third law and run-to-run determinism (no atomics on this path)
    assert np.abs(f_gpu[:, :3].sum(axis=0)).max() < 1e-9 * np.abs(f_gpu[:, :3]).max() * np.sqrt(len(f_gpu))
    f_again = H.gpu_pair_forces("PerturbedLennardJones", pos, (cfg["L"],), nl, params, 3.0, mode="shift", r_list_max=3.4)
    assert np.array_equal(f_again, f_gpu)


# ---------------------------------------------------------------------------
# tile-plan (LDS-staged) kernels: same observables, same tolerance
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["PerturbedLennardJones", "Hertz", "ExpandedYukawa", "Colloid", "DPDConservative"])
@pytest.mark.parametrize("T", [1, 3])
@pytest.mark.parametrize("mode", ["none", "shift", "xplor"])
def test_planned_pair_parity(oracle, name, T, mode):
    a = 1.1 if name != "Colloid" else 1.6
    pos, L, typeid = H.lattice_config(20, a, 0.1 * a, seed=31, ntypes=T)
    box = oracle.make_box(L)
    r_cut = np.full((T, T), 2.5 if name != "Colloid" else 3.2)
    if T > 1:
        r_cut[0, 1] = r_cut[1, 0] = r_cut[0, 0] - 0.3
    r_on = 0.8 * r_cut
    params = _params_table(oracle, name, T)
    nl = oracle.build_nlist(pos, box, r_cut + 0.3, ntypes=T)
    f_ref, v_ref = oracle.pair_forces(name, pos, box, nl, params, r_cut, r_on, mode, ntypes=T, virial=True, nthreads=8)
    info = {}
    rl = float(r_cut.max()) + 0.3
    f_gpu, v_gpu = H.gpu_pair_forces(name, pos, (L,), nl, params, r_cut, r_on, mode, ntypes=T, virial=True, planned=True,
                                     plan_info=info, r_list_max=rl)
    assert info["valid"] == 1, info
    assert_close(f_gpu[:, :3], f_ref[:, :3], what="force")
    assert_close(f_gpu[:, 3], f_ref[:, 3], what="energy")
    assert_close(v_gpu, v_ref, what="virial")
    assert_per_particle(f_gpu, f_ref)
    f2 = H.gpu_pair_forces(name, pos, (L,), nl, params, r_cut, r_on, mode, ntypes=T, planned=True, r_list_max=rl)
    assert np.array_equal(f2, f_gpu)
    # without the hint every pair is re-imaged: same forces to rounding
    f3 = H.gpu_pair_forces(name, pos, (L,), nl, params, r_cut, r_on, mode, ntypes=T, planned=True)
    assert_close(f3, f_gpu, tol=1e-12)


@pytest.mark.parametrize("tpp", [1, 2, 4])
def test_planned_launch_shapes(oracle, tpp):
    """Every lanes-per-particle variant of the tile kernel, N not a multiple of
    the tile size, rows of very different lengths (a slab of vacuum)."""
    cfg = syn.config_plj_sc(22)
    keep = ~((cfg["xyz"][:, 2] > 2.0) & (cfg["xyz"][:, 2] < 6.0))
    pos = syn.pos4(cfg["xyz"][keep][:-7])
    L = cfg["L"]
    box = oracle.make_box(L)
    params = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    nl = oracle.build_nlist(pos, box, 3.4)
    assert nl[0].min() < 0.7 * nl[0].max()
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, nl, params, 3.0, mode="shift", nthreads=8)
    info = {}
    f_gpu = H.gpu_pair_forces("PerturbedLennardJones", pos, (L,), nl, params, 3.0, mode="shift", tpp=tpp, planned=True,
                              plan_info=info, r_list_max=3.4)
    assert info["valid"] == 1 and info["threads_per_particle"] == tpp and info["tile_size"] == 256 // tpp
    assert_close(f_gpu, f_ref)
    assert_per_particle(f_gpu, f_ref)


def test_planned_small_box_and_unsorted_particles(oracle):
    """A box narrower than a tile plus its halo cannot rely on one staged image
    per neighbor: those tiles re-apply the minimum image per pair. Unsorted
    particles make a tile's neighbor set exceed the LDS budget: the plan reports
    invalid and the planned entry point runs the generic kernel. Same answer."""
    cfg = syn.config_plj_sc(9)
    pos = syn.pos4(cfg["xyz"])
    box = oracle.make_box(cfg["L"])
    params = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    nl = oracle.build_nlist(pos, box, 3.4)
    info = {}
    f_gpu = H.gpu_pair_forces("PerturbedLennardJones", pos, (cfg["L"],), nl, params, 3.0, planned=True, plan_info=info,
                              r_list_max=3.4)
    assert info["valid"] == 1
    assert_close(f_gpu, oracle.pair_forces("PerturbedLennardJones", pos, box, nl, params, 3.0))
    # triclinic box through the plan
    tilt = (0.2, -0.1, 0.15)
    cfg10 = syn.config_plj_sc(10)
    pos10 = syn.pos4(cfg10["xyz"])
    nl10 = oracle.build_nlist(pos10, oracle.make_box(cfg10["L"]), 4.5)
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos10, oracle.make_box(cfg10["L"], tilt=tilt), nl10, params, 2.0)
    f_gpu = H.gpu_pair_forces("PerturbedLennardJones", pos10, (cfg10["L"], tilt), nl10, params, 2.0, planned=True,
                              r_list_max=4.5)
    assert_close(f_gpu, f_ref)
    # unsorted particle order => invalid plan, generic kernel, still correct
    cfg = syn.config_plj_sc(20)
    perm = np.argsort(syn.hash64(99, np.arange(8000, dtype=np.uint64), 0))
    pos = syn.pos4(cfg["xyz"][perm])
    box = oracle.make_box(cfg["L"])
    nl = oracle.build_nlist(pos, box, 3.4)
    info = {}
    f_gpu = H.gpu_pair_forces("PerturbedLennardJones", pos, (cfg["L"],), nl, params, 3.0, planned=True, plan_info=info)
    assert info["valid"] == 0 and info["invalid_reason"] == 2
    assert_close(f_gpu, oracle.pair_forces("PerturbedLennardJones", pos, box, nl, params, 3.0, nthreads=8))


def test_planned_ghosts_nonperiodic_and_stale_plan(oracle):
    import ctypes as C

    from azplugins_amd import _lib

    cfg = syn.config_plj_sc(20)
    xyz = cfg["xyz"]
    L = cfg["L"]
    order = np.argsort(xyz[:, 0] > 0.0, kind="stable")
    pos = syn.pos4(xyz[order])
    N = int((xyz[:, 0] <= 0.0).sum())
    box_o = oracle.make_box(L, periodic=(0, 1, 1))
    params = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    nl = oracle.build_nlist(pos, box_o, 2.9, N=N)
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box_o, nl, params, 2.5, N=N)
    info = {}
    f_gpu = H.gpu_pair_forces("PerturbedLennardJones", pos, (L, (0, 0, 0), (0, 1, 1)), nl, params, 2.5, N=N, planned=True,
                              plan_info=info, r_list_max=2.9)
    assert info["valid"] == 1
    assert_close(f_gpu, f_ref)
    # a plan compiled for another list is rejected, not silently used
    a, t = H.gpu_pair_args(pos, (L, (0, 0, 0), (0, 1, 1)), nl, 1, 2.5, 0.0, "none", False, N)
    plan = _lib.PairPlan()
    plan.build(a, H._stream())
    a2, t2 = H.gpu_pair_args(pos, (L, (0, 0, 0), (0, 1, 1)), nl, 1, 2.5, 0.0, "none", False, N)
    p = H._dev(np.atleast_2d(params))
    rc = _lib.lib().azp_pair_forces_planned_perturbed_lennard_jones(plan.handle, C.byref(a2), p.data_ptr(), H._stream())
    assert rc == -1


def test_planned_plj_c2_full_size(oracle):
    """configs[1] at full size through the tile plan."""
    cfg = syn.config_plj_sc(64)
    pos = syn.pos4(cfg["xyz"])
    box = oracle.make_box(cfg["L"])
    params = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    nl = oracle.build_nlist(pos, box, 3.4)
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, nl, params, 3.0, mode="shift", nthreads=16)
    info = {}
    f_gpu = H.gpu_pair_forces("PerturbedLennardJones", pos, (cfg["L"],), nl, params, 3.0, mode="shift", planned=True,
                              plan_info=info, r_list_max=3.4)
    assert info["valid"] == 1
    assert_close(f_gpu, f_ref)
    assert_per_particle(f_gpu, f_ref)
    f_again = H.gpu_pair_forces("PerturbedLennardJones", pos, (cfg["L"],), nl, params, 3.0, mode="shift", planned=True,
                                r_list_max=3.4)
    assert np.array_equal(f_again, f_gpu)


@pytest.mark.parametrize("planned", [False, True])
def test_particle_range_launches(oracle, planned):
    """Sub-range launches (interior first, boundary later, as the domain-decomposed
    step does) reproduce the single full launch; untouched rows keep their values."""
    import ctypes as C

    import torch

    from azplugins_amd import _lib

    cfg = syn.config_plj_sc(20)
    pos = syn.pos4(cfg["xyz"][:-3])
    N = pos.shape[0]
    box = oracle.make_box(cfg["L"])
    params = oracle.pack_pair_params("PerturbedLennardJones", cfg["params"])
    nl = oracle.build_nlist(pos, box, 3.4)
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, nl, params, 3.0, nthreads=8)
    a, t = H.gpu_pair_args(pos, (cfg["L"],), nl, 1, 3.0, 0.0, "none", False, r_list_max=3.4)
    p = H._dev(np.atleast_2d(params))
    lib = _lib.lib()
    if planned:
        plan = _lib.PairPlan()
        plan.build(a, H._stream())

        def launch():
            _lib.check(lib.azp_pair_forces_planned_perturbed_lennard_jones(plan.handle, C.byref(a), p.data_ptr(), H._stream()))
    else:
        def launch():
            _lib.check(lib.azp_pair_forces_perturbed_lennard_jones(C.byref(a), p.data_ptr(), H._stream()))
    split = 5000
    t["force"].fill_(7.0)
    a.range_first, a.range_count = 0, split
    launch()
    torch.cuda.synchronize()
    f1 = t["force"].cpu().numpy().copy()
    assert_close(f1[:split], f_ref[:split])
    tile = 256 if planned else 1
    assert np.all(f1[((split + tile - 1) // tile) * tile:] == 7.0)  # rows beyond the (tile-rounded) range untouched
    a.range_first, a.range_count = split, N - split
    launch()
    torch.cuda.synchronize()
    assert_close(t["force"].cpu().numpy(), f_ref)
    a.range_first, a.range_count = N - 10, 20  # out of bounds
    rc = lib.azp_pair_forces_perturbed_lennard_jones(C.byref(a), p.data_ptr(), H._stream())
    assert rc == -1


# ---------------------------------------------------------------------------
# golden r-sweeps through the kernels: isolated pairs, one per sweep point
# ---------------------------------------------------------------------------
def _isolated_pairs(r_values, gap=40.0):
    """2 n particles: pair k sits at y = k * gap, separated by r_k along x."""
    n = len(r_values)
    xyz = np.zeros((2 * n, 3))
    xyz[0::2, 0] = -0.5 * r_values
    xyz[1::2, 0] = 0.5 * r_values
    xyz[0::2, 1] = xyz[1::2, 1] = (np.arange(n) - 0.5 * n) * gap
    n_neigh = np.ones(2 * n, dtype=np.uint32)
    head = np.arange(2 * n, dtype=np.uint64)
    nlist = np.arange(2 * n, dtype=np.uint32) ^ 1
    L = np.array([200.0, (n + 2) * gap, 200.0])
    return syn.pos4(xyz), L, (n_neigh, head, nlist)


@pytest.mark.parametrize("planned", [False, True])
def test_golden_sweeps(oracle, planned):
    """Every point of tests/golden/sweeps.npz (256 separations x 3 parameter sets x
    {no shift, shift} per evaluator, incl. r = r_wca and r = r_cut exactly) as an
    isolated pair: force_divr * r and pair_eng / 2 per particle."""
    import importlib.util
    import os

    gdir = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(gdir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(os.path.join(gdir, "sweeps.npz"))
    for name, (r_cut, plist) in mg.SWEEPS.items():
        r = gold[name + "_r"]
        pos, L, nl = _isolated_pairs(r)
        for ip, p in enumerate(plist):
            params = oracle.pack_pair_params(name, p)
            for ish, mode in enumerate(("none", "shift")):
                ref = gold[name][ip, ish]  # (evaluated, force_divr, energy)
                f = H.gpu_pair_forces(name, pos, (L, (0, 0, 0), (0, 0, 0)), nl, params, r_cut, mode=mode,
                                      planned=planned, r_list_max=1.06 * r_cut)
                fx_expect = -ref[:, 1] * r  # particle at -r/2: dx = -r
                scale_f = max(np.abs(fx_expect).max(), 1e-300)
                scale_e = max(np.abs(ref[:, 2]).max(), 1e-300)
                assert np.abs(f[0::2, 0] - fx_expect).max() <= 1e-12 * scale_f, (name, ip, mode)
                assert np.abs(f[1::2, 0] + fx_expect).max() <= 1e-12 * scale_f, (name, ip, mode)
                assert np.abs(f[:, 3] - 0.5 * np.repeat(ref[:, 2], 2)).max() <= 1e-12 * scale_e, (name, ip, mode)
                assert not f[:, 1:3].any()
                # not evaluated => exact zeros
                off = np.repeat(ref[:, 0] == 0, 2)
                assert not f[off].any(), (name, ip, mode)


def test_product_nlist_matches_oracle(oracle):
    """The GPU cell-list builder (row N1) lists exactly the oracle's pairs: per
    type pair r_list, bonded exclusions, ghosts-free periodic box."""
    import azplugins_amd as azp

    cfg = syn.config_chains(32, 8, 8, 16)
    n = cfg["xyz"].shape[0]
    typeid = (np.arange(n) // 3) % 2
    snap = azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], typeid=typeid, types=("A", "B"), bonds=cfg["bonds"])
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(snap)
    nl = azp.nlist.Cell(buffer=0.3)
    nl.fused = False   # this test is about the HOOMD-format list (the fused path keeps none: test_gpu_fused_plan.py)
    pot = azp.pair.Hertz(nlist=nl, default_r_cut=2.0)
    pot.r_cut[("A", "B")] = 1.5
    pot.r_cut[("B", "B")] = 2.4
    for pair in (("A", "A"), ("A", "B"), ("B", "B")):
        pot.params[pair] = dict(epsilon=1.0)
    sim.operations.integrator = azp.Integrator(dt=0.001, forces=[pot])
    sim.run(0)
    n_neigh = nl.n_neigh.cpu().numpy().astype(np.int64)
    head = nl.head_list.cpu().numpy()
    nlist = nl.nlist.cpu().numpy()
    # oracle list with the same per-type-pair radii and the bonded exclusions
    rl = np.array([[2.3, 1.8], [1.8, 2.7]])
    n_excl = np.zeros(n, dtype=np.uint32)
    excl = np.zeros((n, 2), dtype=np.uint32)
    for a_, b_ in cfg["bonds"]:
        for me, other in ((a_, b_), (b_, a_)):
            excl[me, n_excl[me]] = other
            n_excl[me] += 1
    pos = syn.pos4(cfg["xyz"], typeid)
    o_n, o_head, o_list = oracle.build_nlist(pos, oracle.make_box(cfg["L"]), rl, ntypes=2, exclusions=(n_excl, excl))
    assert np.array_equal(n_neigh, o_n)
    for i in range(0, n, 37):
        mine = np.sort(nlist[head[i]: head[i] + n_neigh[i]])
        assert np.array_equal(mine, o_list[o_head[i]: o_head[i] + o_n[i]])
    # and the forces through the API (multi-type tables, plan or fallback) match the oracle
    params = np.array([oracle.pack_pair_params("Hertz", dict(epsilon=1.0))] * 4)
    rc = np.array([[2.0, 1.5], [1.5, 2.4]])
    f_ref = oracle.pair_forces("Hertz", pos, oracle.make_box(cfg["L"]), (o_n, o_head, o_list), params, rc, ntypes=2)
    assert_close(np.c_[pot.forces, pot.energies], f_ref)


@pytest.mark.parametrize("mode", ["none", "shift"])
@pytest.mark.parametrize("case", ["lam0", "eps0", "rcut_inside_core", "no_core_pairs", "all_core", "lam1"])
def test_planned_plj_special_cases(oracle, case, mode):
    """The single-type PerturbedLJ fast path (counted energy offsets, tail-only form,
    core-first rows): purely repulsive (lambda = 0), epsilon = 0, a cutoff inside the
    WCA core, no pair / every pair inside the core, lambda = 1 (plain LJ)."""
    pos, L, _ = H.lattice_config(16, 1.1, 0.11, seed=77, ntypes=1)
    box = oracle.make_box(L)
    d = dict(epsilon=1.3, sigma=1.0, attraction_scale_factor=0.5)
    r_cut = 2.5
    if case == "lam0":
        d["attraction_scale_factor"] = 0.0
    elif case == "lam1":
        d["attraction_scale_factor"] = 1.0
    elif case == "eps0":
        d["epsilon"] = 0.0
    elif case == "rcut_inside_core":
        d["sigma"], r_cut = 1.3, 1.4       # r_wca = 1.459 > r_cut
    elif case == "no_core_pairs":
        d["sigma"] = 0.7                   # r_wca = 0.786 < every separation
    elif case == "all_core":
        d["sigma"], r_cut = 2.4, 2.6       # r_wca = 2.69 > r_cut: every listed in-range pair is in the core
    params = oracle.pack_pair_params("PerturbedLennardJones", d)
    nl = oracle.build_nlist(pos, box, r_cut + 0.3)
    f_ref, v_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, nl, params, r_cut, 0.0, mode, virial=True)
    r_wca = 2.0 ** (1.0 / 6.0) * d["sigma"]
    for r_inner in (None, r_wca + 0.3):
        for virial in (False, True):
            info = {}
            out = H.gpu_pair_forces("PerturbedLennardJones", pos, (L,), nl, params, r_cut, 0.0, mode, virial=virial, planned=True,
                                    plan_info=info, r_list_max=r_cut + 0.3, r_inner=r_inner)
            f_gpu = out[0] if virial else out
            assert info["valid"] == 1
            assert_close(f_gpu[:, :3], f_ref[:, :3], what="force")
            assert_close(f_gpu[:, 3], f_ref[:, 3], what="energy")
            if virial:
                assert_close(out[1], v_ref, what="virial")


def test_nlist_single_pass_rebuild_and_overflow(oracle):
    """Rebuilds use fixed-capacity rows and the fill alone; the listed pairs and the
    forces are the same as with exact rows, and a row overflow falls back to exact rows."""
    import azplugins_amd as azp

    cfg = syn.config_plj_sc(14)
    n = cfg["xyz"].shape[0]
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"]))
    nl = azp.nlist.Cell(buffer=0.4)
    nl.fused = False   # the HOOMD-format list and its rebuild protocol
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=2.0)
    pot.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=0.001, forces=[pot])
    sim.run(0)

    def rows():
        nn = nl.n_neigh.cpu().numpy().astype(np.int64)
        hd = nl.head_list.cpu().numpy()
        li = nl.nlist.cpu().numpy()
        return nn, [np.sort(li[hd[i]: hd[i] + nn[i]]) for i in range(n)]

    nn0, rows0 = rows()
    f0 = np.c_[pot.forces, pot.energies]
    assert nl.n_pairs == nn0.sum() and nl.max_neigh == nn0.max()
    pos = syn.pos4(cfg["xyz"])
    o_n, o_head, o_list = oracle.build_nlist(pos, oracle.make_box(cfg["L"]), 2.4)
    assert np.array_equal(nn0, o_n)
    cap = nl._row_capacity
    assert cap >= nn0.max() and cap % 8 == 0
    nl.compute(sim.state, force=True)          # single pass, strided rows
    assert int(nl.head_list[1].item()) == cap and nl.size == n * cap
    nn1, rows1 = rows()
    assert np.array_equal(nn0, nn1) and all(np.array_equal(a, b) for a, b in zip(rows0, rows1))
    pot.compute(0)
    assert_close(np.c_[pot.forces, pot.energies], f0)
    nl._row_capacity = 8                         # guaranteed overflow -> exact rows again
    nl.compute(sim.state, force=True)
    nn2, rows2 = rows()
    assert nl.size == nn0.sum()
    assert np.array_equal(nn0, nn2) and all(np.array_equal(a, b) for a, b in zip(rows0, rows2))
    assert nl._row_capacity == cap


def test_api_virial_and_modes_multitype(oracle):
    """hoomd.azplugins-shaped API: 3 types, xplor mode, virials, PerturbedLJ."""
    import azplugins_amd as azp

    pos, L, typeid = H.lattice_config(18, 1.1, 0.11, seed=5, ntypes=3)
    snap = azp.Snapshot.from_arrays(pos[:, :3], L, typeid=typeid, types=("A", "B", "C"))
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(snap)
    nl = azp.nlist.Cell(buffer=0.3)
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=2.5, default_r_on=2.0, mode="xplor")
    pot.compute_virial = True
    names = ("A", "B", "C")
    tab = H.sym_table(3, PAIR_PARAMS["PerturbedLennardJones"])
    for i in range(3):
        for j in range(i, 3):
            pot.params[(names[i], names[j])] = tab[i][j]
    sim.operations.integrator = azp.Integrator(dt=0.001, forces=[pot])
    sim.run(0)
    params = np.array([oracle.pack_pair_params("PerturbedLennardJones", tab[i][j]) for i in range(3) for j in range(3)])
    box = oracle.make_box(L)
    onl = oracle.build_nlist(pos, box, 2.8, ntypes=3, half=True)
    f_ref, v_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, onl, params, 2.5, 2.0, "xplor", ntypes=3, half=True,
                                      virial=True)
    assert_close(np.c_[pot.forces, pot.energies], f_ref)
    assert_close(pot.virials, v_ref.T)
    assert pot.energy == pytest.approx(f_ref[:, 3].sum(), rel=1e-11)
    assert pot.plan_info["valid"] == 1


@pytest.mark.parametrize("T", [1, 2])
def test_planned_after_particles_moved(oracle, T):
    """The plan is compiled once per neighbor-list build and reused while particles
    move (by up to r_buff / 2 each). The displacement-bounded skipping of buffer
    entries must stay exact: forces at the moved positions, old list, old plan ==
    oracle at the moved positions with the same list. Includes particles that get
    wrapped through the periodic boundary in between."""
    import ctypes as C

    import torch

    from azplugins_amd import _lib

    cfg = syn.config_plj_sc(20)
    n = cfg["xyz"].shape[0]
    L = cfg["L"]
    # shift the lattice so that some sites sit just inside the +x face
    cfg["xyz"] = syn.wrap(cfg["xyz"] + np.array([0.42 * 0.8 ** (-1.0 / 3.0), 0.0, 0.0]), L)
    typeid = (np.arange(n) // 5) % T
    pos0 = syn.pos4(cfg["xyz"], typeid)
    box = oracle.make_box(L)
    r_cut = np.full((T, T), 3.0)
    if T > 1:
        r_cut[0, 1] = r_cut[1, 0] = 2.6
    r_buff = 0.4
    tab = H.sym_table(T, PAIR_PARAMS["PerturbedLennardJones"])
    params = np.array([oracle.pack_pair_params("PerturbedLennardJones", tab[i][j]) for i in range(T) for j in range(T)])
    nl = oracle.build_nlist(pos0, box, r_cut + r_buff, ntypes=T)
    a, t = H.gpu_pair_args(pos0, (L,), nl, T, r_cut, 0.0, "shift", False, r_list_max=3.0 + 2 * r_buff)
    p = H._dev(params)
    lib = _lib.lib()
    plan = _lib.PairPlan()
    plan.build(a, H._stream())
    assert plan.info()["valid"] == 1
    tag = np.arange(n, dtype=np.uint64)
    crossed = 0
    for frac in (0.0, 0.3, 0.99):
        # half of the allowance as a common drift along x (pushes particles through the
        # periodic boundary), half as a random vector
        v = np.stack([syn.normal(77, tag, c) for c in range(3)], axis=1)
        v *= (frac * 0.25 * r_buff * syn.u01(78, tag, 0) / np.linalg.norm(v, axis=1))[:, None]
        v[:, 0] += frac * 0.25 * r_buff
        assert np.linalg.norm(v, axis=1).max() <= 0.5 * r_buff
        xyz = syn.wrap(cfg["xyz"] + v, L)
        crossed += int((np.abs(xyz - cfg["xyz"]) > 0.5 * L).any())
        pos = syn.pos4(xyz, typeid)
        t["pos"].copy_(torch.from_numpy(pos).to("cuda:0"))
        t["force"].fill_(float("nan"))
        _lib.check(lib.azp_pair_forces_planned_perturbed_lennard_jones(plan.handle, C.byref(a), p.data_ptr(), H._stream()))
        torch.cuda.synchronize()
        f_gpu = t["force"].cpu().numpy()
        f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, nl, params, r_cut, 0.0, "shift", ntypes=T, nthreads=8)
        assert_close(f_gpu, f_ref)
        assert_per_particle(f_gpu, f_ref)
        # the generic kernel on the same inputs agrees too
        t["force"].fill_(float("nan"))
        _lib.check(lib.azp_pair_forces_perturbed_lennard_jones(C.byref(a), p.data_ptr(), H._stream()))
        torch.cuda.synchronize()
        assert_close(t["force"].cpu().numpy(), f_ref)
    assert crossed >= 1  # some particle was wrapped through the box between build and use


@pytest.mark.parametrize("frac", [0.0, 0.2, 0.45, 0.9])
def test_planned_displacement_bound_is_exact(oracle, frac):
    """azp_pair_args.displacement_bound: after the plan is built every particle moves
    by at most frac * r_buff / 2; the planned kernel, told that bound, may stop its rows
    before buffer entries that cannot have come into range -- the forces must equal the
    oracle's on the moved positions with the old list. frac = 0: all buffer entries
    skipped; 0.2, 0.45: the outer half of the buffer shell; 0.9: nothing skipped."""
    r_cut, r_buff = 2.5, 0.4
    pos, L, _ = H.lattice_config(16, 1.1, 0.11, seed=91, ntypes=1)
    box = oracle.make_box(L)
    params = oracle.pack_pair_params("PerturbedLennardJones", PAIR_PARAMS["PerturbedLennardJones"](0, 0))
    nl = oracle.build_nlist(pos, box, r_cut + r_buff)
    n = pos.shape[0]
    tag = np.arange(n, dtype=np.uint64)
    v = np.stack([syn.normal(3, tag, c) for c in range(3)], axis=1)
    amp = frac * 0.5 * r_buff
    v *= (amp * syn.u01(3, tag, 9) ** (1.0 / 3.0) / np.linalg.norm(v, axis=1))[:, None]
    # half of the particles sit exactly on the bound
    v[::2] *= (amp / np.maximum(np.linalg.norm(v[::2], axis=1), 1e-300))[:, None]
    moved = pos.copy()
    moved[:, :3] = syn.wrap(pos[:, :3] + v, L)
    f_ref = oracle.pair_forces("PerturbedLennardJones", moved, box, nl, params, r_cut, 0.0, "shift")

    a, t = H.gpu_pair_args(pos, (L,), nl, 1, r_cut, 0.0, "shift", False, None, 0, 0, r_cut + 2 * r_buff)
    p = H._dev(np.atleast_2d(params).astype(np.float64))
    plan = H._lib.PairPlan()
    plan.build(a, H._stream())                      # built on the ORIGINAL positions
    assert plan.info()["valid"] == 1
    import torch
    t["pos"].copy_(torch.from_numpy(moved))          # then the particles move
    results = {}
    for label, known, bound in (("unknown", 0, 0.0), ("bound", 1, amp * (1 + 1e-12))):
        a.has_displacement_bound, a.displacement_bound = known, bound
        H._lib.check(H._lib.lib().azp_pair_forces_planned_perturbed_lennard_jones(plan.handle, H.C.byref(a), p.data_ptr(),
                                                                                 H._stream()), "planned")
        results[label] = H._finish(t, False)
    assert_close(results["unknown"], f_ref)
    assert_close(results["bound"], f_ref)


def test_north_star_full_size_properties():
    """N = 1,048,576 (BASELINE.json's headline size), through size-independent
    properties: total force = 0 (third law, though the kernel uses full lists); the
    planned kernel agrees with the generic one; stopping rows at the displacement bound
    changes no bit (skipped pairs are out of range and would have added +0.0)."""
    import azplugins_amd as azp

    cfg = syn.config_north_star(64)
    n = cfg["xyz"].shape[0]
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"]))
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=cfg["r_cut"], mode="shift")
    pot.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot])
    sim.run(0)
    assert pot.plan_info["valid"] == 1 and n == 1048576
    f_bound = pot.force_tensor.clone()
    pot.use_displacement_bound = False
    pot.compute(0)
    f_whole = pot.force_tensor.clone()
    assert bool((f_bound == f_whole).all())
    pot.use_plan = False
    pot.compute(0)
    f_generic = pot.force_tensor.clone()
    scale = float(f_generic[:, :3].abs().max())
    assert float((f_generic - f_whole).abs().max()) <= 1e-11 * max(scale, float(f_generic[:, 3].abs().max()))
    total = f_whole[:, :3].sum(dim=0).abs().max()
    assert float(total) <= 1e-9 * scale
    assert abs(nl.n_pairs / n - 136.26) < 0.05


def test_api_r_cut_and_mode_change_rebuild_list_and_plan(oracle):
    """Changing r_cut (within and beyond the buffer) or the mode after the first compute:
    HOOMD rebuilds the neighbor list when a consumer's r_cut matrix changes; here the list
    and the tile plan (whose row classes were cut against the old r_cut) must follow, or
    pairs between the old and the new cutoff are silently dropped."""
    import azplugins_amd as azp

    pos, L, _ = H.lattice_config(14, 1.1, 0.11, seed=33, ntypes=1)
    p = PAIR_PARAMS["PerturbedLennardJones"](0, 0)
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(pos[:, :3], L))
    nl = azp.nlist.Cell(buffer=0.3)
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=2.0, mode="none")
    pot.params[("A", "A")] = p
    sim.operations.integrator = azp.Integrator(dt=0.001, forces=[pot])
    sim.run(0)
    box = oracle.make_box(L)
    params = oracle.pack_pair_params("PerturbedLennardJones", p)
    builds = nl.num_builds
    for r_cut, mode in ((2.0, "none"), (2.2, "none"), (2.2, "shift"), (2.9, "shift"), (1.6, "xplor")):
        pot.r_cut[("A", "A")] = r_cut
        pot.mode = mode
        pot.r_on[("A", "A")] = 0.8 * r_cut
        pot.compute(0)
        onl = oracle.build_nlist(pos, box, r_cut + 0.3, half=True)
        f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, onl, params, r_cut, 0.8 * r_cut, mode, half=True)
        assert_close(np.c_[pot.forces, pot.energies], f_ref, what="r_cut=%g mode=%s" % (r_cut, mode))
        assert pot.plan_info["valid"] == 1
    assert nl.num_builds >= builds + 3  # every r_cut change rebuilt the list


def test_api_parameter_change_between_steps_of_a_run(oracle):
    """The per-step path queues the force kernel behind the list's distance check with the argument struct of the
    previous step (pair.py, _compute_speculative): new parameters, a new mode or a virial request between two steps must
    reach the very next launch, not the one after the next list rebuild."""
    import azplugins_amd as azp

    pos, L, _ = H.lattice_config(12, 1.1, 0.11, seed=34, ntypes=1)
    p = dict(PAIR_PARAMS["PerturbedLennardJones"](0, 0))
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(pos[:, :3], L))
    nl = azp.nlist.Cell(buffer=0.4)
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=2.5, mode="none")
    pot.params[("A", "A")] = p
    sim.operations.integrator = azp.Integrator(dt=0.0005, forces=[pot], methods=[azp.ConstantVolume()])
    sim.operations.tuners.clear()
    sim.run(3)
    box = oracle.make_box(L)
    builds = nl.num_builds
    for change in ("epsilon", "mode", "virial", "lambda"):
        if change == "epsilon":
            p["epsilon"] = 1.7 * p["epsilon"]
        elif change == "lambda":
            p["attraction_scale_factor"] = 0.35
        if change in ("epsilon", "lambda"):
            pot.params[("A", "A")] = p
        if change == "mode":
            pot.mode = "shift"
        if change == "virial":
            pot.compute_virial = True
        sim.run(2)
        x = syn.pos4(sim.state.pos[:, :3].cpu().numpy())
        onl = oracle.build_nlist(x, box, 2.9, half=True)
        f_ref = oracle.pair_forces("PerturbedLennardJones", x, box, onl, oracle.pack_pair_params("PerturbedLennardJones", p), 2.5,
                                   mode=pot.mode, half=True)
        assert_close(np.c_[pot.forces, pot.energies], f_ref, what="after changing %s" % change)
    assert nl.num_builds == builds  # (nothing here asked for a new list: the launches above took the per-step path)


@pytest.mark.parametrize("T", [1, 2])
def test_planned_dpd_parity(oracle, T):
    """Tile-staged DPD thermostat kernel (azp_dpd_forces_planned_general_weight: positions,
    velocities and tags staged in LDS) against the oracle with the same Philox stream, with
    and without the r_list_max hint, virial included; and with particles moved after the
    plan was built (displacement bound: rows stop before the buffer shells that cannot have
    come into range -- exact)."""
    cfg = syn.config_dpd(6000)
    n = cfg["xyz"].shape[0]
    typeid = (np.arange(n) % T) if T > 1 else None
    pos = syn.pos4(cfg["xyz"], typeid)
    vel = np.zeros((n, 4))
    vel[:, :3] = cfg["vel"]
    vel[:, 3] = 1.0
    box = oracle.make_box(cfg["L"])
    tab = H.sym_table(T, lambda i, j: dict(A=25.0 - 5 * (i + j), gamma=4.5 + i + j, s=[0.5, 1.0, 2.0][i + j]))
    params = np.array([oracle.pack_pair_params("DPDGeneralWeight", tab[i][j]) for i in range(T) for j in range(T)])
    r_cut = np.full((T, T), 1.0)
    if T > 1:
        r_cut[0, 1] = r_cut[1, 0] = 0.9
    nl_f = oracle.build_nlist(pos, box, r_cut + 0.4, ntypes=T)
    kw = dict(kT=1.0, dt=0.01, seed=7, timestep=424242, ntypes=T)
    f_ref, v_ref = oracle.dpd_forces(pos, vel, cfg["tag"], box, nl_f, params, r_cut, virial=True, **kw)
    for hint in (0.0, 1.0 + 2 * 0.4):
        info = {}
        f_gpu, v_gpu = H.gpu_dpd_forces(pos, vel, cfg["tag"], (cfg["L"],), nl_f, params, r_cut, virial=True, planned=True, plan_info=info,
                                        r_list_max=hint, **kw)
        assert info["valid"] == 1
        assert_close(f_gpu, f_ref, what="planned dpd force")
        assert_close(v_gpu, v_ref, what="planned dpd virial")
        assert_per_particle(f_gpu, f_ref)
    # bound 0: every buffer entry skipped, same answer
    f0 = H.gpu_dpd_forces(pos, vel, cfg["tag"], (cfg["L"],), nl_f, params, r_cut, planned=True, r_list_max=1.8, displacement_bound=0.0, **kw)
    assert_close(f0, f_ref, what="planned dpd bound 0")


def test_planned_dpd_after_particles_moved(oracle):
    import ctypes as C

    import torch

    from azplugins_amd import _lib

    cfg = syn.config_dpd(6000)
    n = cfg["xyz"].shape[0]
    pos0 = syn.pos4(cfg["xyz"])
    vel = np.zeros((n, 4))
    vel[:, :3] = cfg["vel"]
    vel[:, 3] = 1.0
    box = oracle.make_box(cfg["L"])
    params = np.atleast_2d(oracle.pack_pair_params("DPDGeneralWeight", cfg["params"]))
    r_buff = 0.4
    nl = oracle.build_nlist(pos0, box, 1.0 + r_buff)
    a, t = H.gpu_pair_args(pos0, (cfg["L"],), nl, 1, 1.0, 0.0, "none", False, r_list_max=1.0 + 2 * r_buff)
    p = H._dev(params)
    v = H._dev(vel)
    tg = H._dev(cfg["tag"], np.uint32)
    plan = _lib.PairPlan()
    plan.build(a, H._stream())
    assert plan.info()["valid"] == 1
    tag = np.arange(n, dtype=np.uint64)
    for frac in (0.0, 0.3, 0.7, 0.99):
        d = np.stack([syn.normal(5, tag, c) for c in range(3)], axis=1)
        amp = frac * 0.5 * r_buff
        d *= (amp / np.linalg.norm(d, axis=1))[:, None]  # every particle sits ON the bound
        moved = pos0.copy()
        moved[:, :3] = syn.wrap(pos0[:, :3] + d, cfg["L"])
        t["pos"].copy_(torch.from_numpy(moved))
        f_ref = oracle.dpd_forces(moved, vel, cfg["tag"], box, nl, params, 1.0, 1.0, 0.01, 7, 99)
        for known, bound in ((0, 0.0), (1, amp * (1 + 1e-12))):
            a.has_displacement_bound, a.displacement_bound = known, bound
            dd = _lib.DPDArgs()
            dd.pair = a
            dd.d_vel, dd.d_tag = v.data_ptr(), tg.data_ptr()
            dd.timestep, dd.deltaT, dd.T, dd.seed = 99, 0.01, 1.0, 7
            t["force"].fill_(float("nan"))
            _lib.check(_lib.lib().azp_dpd_forces_planned_general_weight(plan.handle, C.byref(dd), p.data_ptr(), H._stream()), "planned dpd")
            torch.cuda.synchronize()
            assert_close(t["force"].cpu().numpy(), f_ref, what="frac %g known %d" % (frac, known))


@pytest.mark.parametrize("T,mode", [(1, "none"), (1, "shift"), (2, "shift")])
def test_planned_aniso_parity(oracle, T, mode):
    """Tile-staged TwoPatchMorse kernel (patch directors of the staged neighbors computed once
    per tile): forces, torques, energies, virial against the oracle."""
    cfg = syn.config_tpm(12, 12, 16)
    n = cfg["xyz"].shape[0]
    typeid = (np.arange(n) % T) if T > 1 else None
    pos = syn.pos4(cfg["xyz"], typeid)
    box = oracle.make_box(cfg["L"])
    tab = H.sym_table(T, lambda i, j: dict(cfg["params"], M_d=1.8341 + 0.2 * (i + j), repulsion=bool((i + j) % 2)))
    params = np.array([oracle.pack_pair_params("TwoPatchMorse", tab[i][j]) for i in range(T) for j in range(T)])
    nl_f = oracle.build_nlist(pos, box, 2.0, ntypes=T)
    f_ref, t_ref, v_ref = oracle.aniso_forces_tpm(pos, cfg["orientation"], box, nl_f, params, 1.6, mode, ntypes=T, virial=True)
    for hint, bound in ((0.0, None), (1.6 + 0.8, None), (1.6 + 0.8, 0.0)):
        info = {}
        f_gpu, t_gpu, v_gpu = H.gpu_aniso_forces(pos, cfg["orientation"], (cfg["L"],), nl_f, params, 1.6, mode, ntypes=T, virial=True,
                                                 planned=True, plan_info=info, r_list_max=hint, displacement_bound=bound)
        assert info["valid"] == 1
        assert_close(f_gpu[:, :3], f_ref[:, :3], what="planned aniso force")
        assert_close(f_gpu[:, 3], f_ref[:, 3], what="planned aniso energy")
        assert_close(t_gpu[:, :3], t_ref[:, :3], what="planned aniso torque")
        assert_close(v_gpu, v_ref, what="planned aniso virial")
        assert not t_gpu[:, 3].any()
