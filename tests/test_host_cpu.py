"""CPU-only checks (no GPU): oracle self-consistency, synthetic-data hash
agreement between numpy and C, host-side logic of the Python mirror, and that
libazp.so loads and exports every symbol include/azp.h declares."""

import ctypes as C
import os
import re

import numpy as np
import pytest

import azplugins_amd as azp
from azplugins_amd import _lib
from azplugins_amd import synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "azp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(azp_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libazp.so does not export %s" % name
    # and the ctypes binding covers exactly the declared set
    assert declared == set(_lib.SYMBOLS)


def test_abi_struct_sizes_match_header():
    """Compile a tiny C program against include/azp.h and compare sizeof()."""
    import subprocess
    import tempfile

    names = ["azp_box", "azp_pair_args", "azp_dpd_args", "azp_aniso_args", "azp_bond_args", "azp_cell_grid",
             "azp_nlist_args", "azp_plj_params", "azp_hertz_params", "azp_yukawa_params", "azp_colloid_params",
             "azp_dpd_params", "azp_tpm_params", "azp_dw_params", "azp_quartic_params", "azp_bond_entry",
             "azp_barrier_args", "azp_nve_args"]
    src = '#include <stdio.h>\n#include "azp.h"\nint main(){' + "".join(
        'printf("%%zu\\n", sizeof(%s));' % n for n in names) + "return 0;}"
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "s.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "s.c"), "-o",
                               os.path.join(d, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "s")]).split()]
    got = dict(zip(names, sizes))
    assert got["azp_box"] == C.sizeof(_lib.Box)
    assert got["azp_pair_args"] == C.sizeof(_lib.PairArgs)
    assert got["azp_dpd_args"] == C.sizeof(_lib.DPDArgs)
    assert got["azp_aniso_args"] == C.sizeof(_lib.AnisoArgs)
    assert got["azp_bond_args"] == C.sizeof(_lib.BondArgs)
    assert got["azp_cell_grid"] == C.sizeof(_lib.CellGrid)
    assert got["azp_nlist_args"] == C.sizeof(_lib.NlistArgs)
    # parameter structs are byte-compatible with the reference's param_type
    assert [got[n] for n in names[7:15]] == [32, 8, 32, 32, 32, 48, 32, 64]
    assert got["azp_bond_entry"] == 8
    assert got["azp_barrier_args"] == C.sizeof(_lib.BarrierArgs)
    assert got["azp_nve_args"] == C.sizeof(_lib.NVEArgs)


def test_oracle_struct_sizes(oracle):
    l = oracle.lib()
    assert [l.azo_sizeof(i) for i in range(8)] == [32, 8, 32, 32, 32, 48, 32, 64]
    assert l.azo_sizeof(10) == C.sizeof(oracle.Box)
    assert l.azo_sizeof(11) == C.sizeof(oracle.PairArgs)
    assert l.azo_sizeof(12) == C.sizeof(oracle.DPDArgs)
    assert l.azo_sizeof(13) == C.sizeof(oracle.AnisoArgs)
    assert l.azo_sizeof(14) == C.sizeof(oracle.BondArgs)


def test_hash_rng_numpy_matches_c(oracle):
    tags = np.array([0, 1, 2, 12345, 2**32 + 5, 2**63 - 1], dtype=np.uint64)
    for seed in (0, 1, 7, 2**40 + 3):
        for comp in (0, 1, 2, 64):
            h = syn.hash64(seed, tags, comp)
            for t, v in zip(tags, h):
                assert int(v) == oracle.hash64(seed, int(t), comp)
    u = syn.u01(5, np.arange(1000, dtype=np.uint64), 2)
    assert np.array_equal(u, oracle.u01_array(5, 0, 1000, 2))
    assert 0.0 <= u.min() and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.03


def test_param_round_trip_host():
    """Param dict -> C struct -> dict, as the reference asserts after attach
    (src/pytest/test_pair.py:349, test_bond.py:223)."""
    cases = [
        (azp.pair.PerturbedLennardJones, dict(epsilon=2.0, sigma=1.05, attraction_scale_factor=0.5)),
        (azp.pair.PerturbedLennardJones, dict(epsilon=2.0, sigma=0.85, attraction_scale_factor=0.0)),
        (azp.pair.Colloid, dict(A=100.0, a_1=1.5, a_2=0.75, sigma=1.05)),
        (azp.pair.Hertz, dict(epsilon=2.0)),
        (azp.pair.ExpandedYukawa, dict(epsilon=1.0, kappa=3.0, delta=1.0)),
    ]
    for cls, d in cases:
        p = cls(nlist=azp.nlist.Cell(buffer=0.4), default_r_cut=3.0)
        back = p._unpack(p._pack(d))
        assert back.keys() == d.keys()
        for k in d:
            assert back[k] == pytest.approx(d[k], rel=1e-15)
    t = azp.pair.TwoPatchMorse(nlist=azp.nlist.Cell(buffer=0.4), default_r_cut=1.6)
    d = dict(M_d=1.8341, M_r=0.0302, r_eq=1.0043, omega=5.0, alpha=0.40, repulsion=False)
    back = t._unpack(t._pack(d))
    assert np.allclose([back[k] for k in d], list(d.values()))
    q = azp.bond.Quartic()
    d = dict(k=1434.3, r_0=1.5, b_1=-0.7589, b_2=0.0, U_0=67.2234, sigma=1.0, epsilon=1.0, delta=0.5)
    assert q._unpack(q._pack(d)) == d
    w = azp.bond.DoubleWell()
    d = dict(r_0=1.0, r_1=2.0, U_1=1.0, U_tilt=0.5)
    assert w._unpack(w._pack(d)) == d


def test_param_structs_match_oracle(oracle):
    """Product host-side packing == oracle packing (both restate the reference's
    dict constructors)."""
    nl = azp.nlist.Cell(buffer=0.4)
    d = dict(epsilon=1.3, sigma=0.97, attraction_scale_factor=0.25)
    assert np.array_equal(azp.pair.PerturbedLennardJones(nl, 3.0)._pack(d), oracle.pack_pair_params("PerturbedLennardJones", d))
    d = dict(A=10.0, a_1=1.5, a_2=0.0, sigma=1.1)
    assert np.array_equal(azp.pair.Colloid(nl, 3.0)._pack(d), oracle.pack_pair_params("Colloid", d))
    d = dict(M_d=1.8, M_r=0.03, r_eq=1.0, omega=5.0, alpha=0.4, repulsion=True)
    assert np.array_equal(azp.pair.TwoPatchMorse(nl, 1.6)._pack(d).view(np.uint8), oracle.pack_pair_params("TwoPatchMorse", d).view(np.uint8))
    d = dict(k=3.0, r_0=1.5, b_1=-0.7, b_2=0.1, U_0=2.0, sigma=0.9, epsilon=1.1, delta=0.2)
    assert np.array_equal(azp.bond.Quartic()._pack(d), oracle.pack_bond_params("Quartic", d))
    d = dict(r_0=1.0, r_1=2.0, U_1=1.0, U_tilt=0.5)
    assert np.array_equal(azp.bond.DoubleWell()._pack(d), oracle.pack_bond_params("DoubleWell", d))


def test_type_parameter_validation():
    p = azp.pair.Hertz(nlist=azp.nlist.Cell(buffer=0.4), default_r_cut=1.0)
    p.params[("A", "B")] = dict(epsilon=2)
    assert p.params[("B", "A")] == dict(epsilon=2.0)
    with pytest.raises(ValueError):
        p.params[("A", "A")] = dict(epsilon=1.0, sigma=1.0)
    with pytest.raises(ValueError):
        p.params[("A", "A")] = dict()
    with pytest.raises(KeyError):
        p.params["A"] = dict(epsilon=1.0)
    assert p.r_cut[("A", "A")] == 1.0
    p.r_cut[("A", "B")] = 2.5
    assert p.r_cut[("B", "A")] == 2.5
    # accepted modes follow the reference (src/pair.py:108, :215)
    with pytest.raises(ValueError):
        azp.pair.Hertz(nlist=azp.nlist.Cell(buffer=0.4), default_r_cut=1.0, mode="bogus")
    d = azp.pair.DPDGeneralWeight(nlist=azp.nlist.Cell(buffer=0.4), kT=1.0, default_r_cut=1.0)
    assert d.mode == "none"
    with pytest.raises(ValueError):
        d.mode = "shift"
    q = azp.bond.Quartic()
    q.params["A-A"] = dict(k=1.0, r_0=1.5, b_1=0.0, b_2=0.0, U_0=0.0, sigma=1.0, epsilon=1.0)
    assert q.params["A-A"]["delta"] == 0.0  # default (src/bond.py:153)


def test_external_barrier_host_logic(oracle):
    """hoomd.azplugins.external mirror: parameter dict semantics of the reference's
    test_create (src/pytest/test_external.py:36-92), location variants, class names,
    and the evaluators' validity rules through the C ABI (host functions, no GPU)."""
    import ctypes as C

    for cls, name in ((azp.external.PlanarHarmonicBarrier, "PlanarHarmonicBarrier"),
                      (azp.external.SphericalHarmonicBarrier, "SphericalHarmonicBarrier")):
        b = cls(location=3.0)
        assert b._cpp_class_name == name
        b.params["A"].update(dict(k=10.0, offset=0.5))
        assert b.params["A"] == dict(k=10.0, offset=0.5)
        b.params["A"]["k"] = 20
        assert b.params["A"] == dict(k=20.0, offset=0.5)
        with pytest.raises(ValueError):
            b.params["B"] = dict(k=1.0)
        with pytest.raises(ValueError):
            b.params["B"] = dict(k=1.0, offset=0.0, extra=1.0)
        assert b._location_at(0) == 3.0
        b.location = lambda timestep: 5.0 if timestep <= 1 else 4.0
        assert b._location_at(1) == 5.0 and b._location_at(2) == 4.0
        with pytest.raises(azp.AzpError):
            b.forces  # not attached
    l = _lib.lib()
    box = _lib.make_box((20.0, 20.0, 20.0))
    # planar: -L/2 <= H < L/2 (src/PlanarBarrierEvaluator.h:50-58); spherical: 0 <= R and 2R <= min L
    assert l.azp_planar_barrier_valid(9.9, C.byref(box)) == 1
    assert l.azp_planar_barrier_valid(10.0, C.byref(box)) == 0
    assert l.azp_planar_barrier_valid(-10.0, C.byref(box)) == 1
    assert l.azp_planar_barrier_valid(-10.1, C.byref(box)) == 0
    assert l.azp_spherical_barrier_valid(10.0, C.byref(box)) == 1
    assert l.azp_spherical_barrier_valid(10.1, C.byref(box)) == 0
    assert l.azp_spherical_barrier_valid(-0.1, C.byref(box)) == 0
    # triclinic: makeCoordinates shears the corners, lo.y = -(Ly/2 + yz Lz/2) (HOOMD BoxDim), so
    # the accepted range of H widens with yz > 0 and narrows with yz < 0
    tri = _lib.make_box((20.0, 20.0, 10.0), tilt=(0.1, 0.0, 0.4))
    assert l.azp_planar_barrier_valid(11.9, C.byref(tri)) == 1
    assert l.azp_planar_barrier_valid(12.0, C.byref(tri)) == 0
    assert l.azp_planar_barrier_valid(-12.0, C.byref(tri)) == 1
    assert l.azp_planar_barrier_valid(-12.1, C.byref(tri)) == 0
    tri = _lib.make_box((20.0, 20.0, 10.0), tilt=(0.0, 0.0, -0.4))
    assert l.azp_planar_barrier_valid(7.9, C.byref(tri)) == 1
    assert l.azp_planar_barrier_valid(8.0, C.byref(tri)) == 0


def test_cpp_class_names_match_reference_module():
    """The names the reference registers in _azplugins (src/module.cc:114-164 via
    the export_*.cc.inc templates)."""
    assert azp.pair.Colloid._cpp_class_name == "PotentialPairColloid"
    assert azp.pair.ExpandedYukawa._cpp_class_name == "PotentialPairExpandedYukawa"
    assert azp.pair.Hertz._cpp_class_name == "PotentialPairHertz"
    assert azp.pair.PerturbedLennardJones._cpp_class_name == "PotentialPairPerturbedLennardJones"
    assert azp.pair.DPDGeneralWeight._cpp_class_name == "PotentialPairDPDThermoGeneralWeight"
    assert azp.pair.DPDConservativeGeneralWeight._cpp_class_name == "PotentialPairConservativeGeneralWeight"
    assert azp.pair.TwoPatchMorse._cpp_class_name == "AnisoPotentialPairTwoPatchMorse"
    assert azp.bond.DoubleWell._cpp_class_name == "PotentialBondDoubleWell"
    assert azp.bond.Quartic._cpp_class_name == "PotentialBondQuartic"


def test_product_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sim = azp.Simulation(device="cpu")
    with pytest.raises(azp.AzpError):
        sim.create_state_from_snapshot(azp.two_particle_snapshot())


def test_oracle_half_equals_full(oracle):
    """Third-law scatter over a half list == per-particle gather over a full
    list (same pairs, different summation order)."""
    pos, L, typeid = __import__("helpers").lattice_config(8, 1.1, 0.12, seed=11, ntypes=2)
    box = oracle.make_box(L)
    T = 2
    params = np.array([oracle.pack_pair_params("PerturbedLennardJones",
                                               dict(epsilon=1.0 + 0.1 * (i + j), sigma=1.0, attraction_scale_factor=0.5))
                       for i in range(T) for j in range(T)])
    r_cut = 2.5
    nl_h = oracle.build_nlist(pos, box, r_cut + 0.3, ntypes=T, half=True)
    nl_f = oracle.build_nlist(pos, box, r_cut + 0.3, ntypes=T, half=False)
    assert nl_f[0].sum() == 2 * nl_h[0].sum()
    for mode in ("none", "shift", "xplor"):
        fh, vh = oracle.pair_forces("PerturbedLennardJones", pos, box, nl_h, params, r_cut, r_on=2.0, mode=mode, ntypes=T,
                                    half=True, virial=True)
        ff, vf = oracle.pair_forces("PerturbedLennardJones", pos, box, nl_f, params, r_cut, r_on=2.0, mode=mode, ntypes=T,
                                    half=False, virial=True)
        fo = oracle.pair_forces("PerturbedLennardJones", pos, box, nl_f, params, r_cut, r_on=2.0, mode=mode, ntypes=T,
                                nthreads=2)
        assert np.allclose(fh, ff, rtol=0, atol=1e-11 * np.abs(ff).max())
        assert np.allclose(vh, vf, rtol=0, atol=1e-11 * np.abs(vf).max())
        assert np.array_equal(fo, ff)
        # Newton's third law: net force vanishes
        assert np.abs(ff[:, :3].sum(axis=0)).max() < 1e-9 * np.abs(ff[:, :3]).max()


def test_oracle_min_image_forms_agree(oracle):
    """Compare form (orthorhombic) and rint form (used for triclinic boxes) give
    the same image away from ties."""
    rng = syn.u01(3, np.arange(3000, dtype=np.uint64), 0).reshape(-1, 3)
    L = np.array([7.0, 9.0, 11.0])
    b_ortho = oracle.make_box(L)
    b_tric = oracle.make_box(L, tilt=(1e-300, 0.0, 0.0))  # forces the rint path, numerically orthorhombic
    for row in rng:
        w = (row - 0.5) * 2.9 * L
        a = w.copy()
        b = w.copy()
        oracle.lib().azo_min_image(C.byref(b_ortho), a.ctypes.data_as(C.POINTER(C.c_double)))
        oracle.lib().azo_min_image(C.byref(b_tric), b.ctypes.data_as(C.POINTER(C.c_double)))
        if np.all(np.abs(np.abs(w / L) % 1.0 - 0.5) > 1e-9) and np.all(np.abs(w) < 1.5 * L):
            assert np.allclose(a, b, atol=1e-12)
            assert np.all(np.abs(a) <= 0.5 * L + 1e-12)


def test_synthetic_configs_shapes():
    c = syn.config_c1()
    assert c["xyz"].shape == (4096, 3) and np.all(np.abs(c["xyz"]) <= 8.0)
    c = syn.config_plj_sc(8)
    assert c["xyz"].shape == (512, 3)
    rho = 512 / np.prod(c["L"])
    assert rho == pytest.approx(0.8, rel=1e-12)
    c = syn.config_north_star(4)
    assert c["xyz"].shape == (256, 3) and 256 / np.prod(c["L"]) == pytest.approx(0.8, rel=1e-12)
    c = syn.config_chains(32, 4, 4, 32)
    assert c["bonds"].shape == (31 * 16, 2)
    d = c["xyz"][c["bonds"][:, 1]] - c["xyz"][c["bonds"][:, 0]]
    d -= c["L"] * np.round(d / c["L"])
    assert np.all(np.linalg.norm(d, axis=1) < 1.4)
    c = syn.config_tpm(4, 4, 8)
    assert np.allclose(np.linalg.norm(c["orientation"], axis=1), 1.0)
    c = syn.config_dpd(4096)
    assert c["vel"].shape == (4096, 3) and abs(c["vel"].var() - 1.0) < 0.1
    assert sorted(c["tag"].tolist()) == list(range(4096))


def test_constant_volume_accepts_only_all_particles():
    """The NVE kernels integrate all N particles: a filter other than All() is an
    error, not silently ignored (hoomd.md.methods.ConstantVolume(filter=...))."""
    assert isinstance(azp.ConstantVolume().filter, azp.All)
    assert azp.ConstantVolume(filter=azp.All()).filter == azp.All()
    with pytest.raises(azp.AzpError):
        azp.ConstantVolume(filter=[0, 1, 2])


def test_azplugins_extension_module_classes_and_dict_round_trip(oracle):
    """The compiled _azplugins module (pybind11, HOOMD-free): exactly the class names the
    reference registers for the force path (src/module.cc:114-164: base name on the CPU,
    + "GPU" under ENABLE_HIP) and the C++ dict -> param_type -> dict round trip, which the
    reference asserts with == after attaching (src/pytest/test_pair.py:349,
    src/pytest/test_bond.py:223) for sigma = 1.05, 0.5, 0.85, 1.0."""
    m = _lib.ext_module()
    names = {n for n in dir(m) if "Potential" in n}
    base = ["AnisoPotentialPairTwoPatchMorse", "PotentialPairColloid", "PotentialPairExpandedYukawa", "PotentialPairHertz",
            "PotentialPairPerturbedLennardJones", "PotentialPairDPDThermoGeneralWeight", "PotentialBondDoubleWell",
            "PotentialBondQuartic"]
    assert names == set(base) | {b + "GPU" for b in base} | {"PotentialPairConservativeGeneralWeight"}
    assert m.PotentialPairHertzGPU.on_gpu and not m.PotentialPairHertz.on_gpu
    # param_type sizes (byte-compatible with the reference's structs)
    assert [getattr(m, n).param_size for n in ("PotentialPairPerturbedLennardJones", "PotentialPairHertz", "PotentialPairExpandedYukawa",
                                               "PotentialPairColloid", "PotentialPairDPDThermoGeneralWeight",
                                               "AnisoPotentialPairTwoPatchMorse", "PotentialBondDoubleWell", "PotentialBondQuartic")] \
        == [32, 8, 32, 32, 32, 48, 32, 64]
    plj = m.PotentialPairPerturbedLennardJonesGPU(["A", "B"])
    for sigma in (1.05, 0.5, 0.85, 1.0):  # the values of the reference's test cases
        d = dict(epsilon=2.0, sigma=sigma, attraction_scale_factor=0.5)
        plj.setParams("A", "B", d)
        assert plj.getParams("A", "B") == d and plj.getParams("B", "A") == d  # exact, both orderings
    assert not plj.hasParams("A", "A") and plj.hasParams("B", "A")
    raw = np.frombuffer(plj.params_bytes(), dtype=np.float64).reshape(4, 4)
    assert np.array_equal(raw[1], oracle.pack_pair_params("PerturbedLennardJones", d)) and np.array_equal(raw[1], raw[2])
    assert not raw[0].any()
    plj.setRCut("A", "B", 3.0)
    plj.setROn("B", "A", 2.0)
    assert plj.getRCut("B", "A") == 3.0 and plj.rcutsq() == [0.0, 9.0, 9.0, 0.0] and plj.ronsq()[1] == 4.0
    plj.mode = "xplor"
    assert plj.mode == "xplor" and plj.shift_mode == 2
    with pytest.raises(RuntimeError):
        plj.mode = "bogus"
    with pytest.raises(RuntimeError):
        plj.setParams("A", "C", d)  # unknown type
    with pytest.raises(KeyError):
        plj.setParams("A", "A", dict(epsilon=1.0, sigma=1.0))  # missing key, as the reference's dict constructor
    cases = [
        ("PotentialPairColloid", dict(A=100.0, a_1=1.5, a_2=0.75, sigma=1.05)),
        ("PotentialPairExpandedYukawa", dict(epsilon=1.0, kappa=3.0, delta=1.0)),
        ("PotentialPairHertz", dict(epsilon=2.0)),
        ("PotentialPairDPDThermoGeneralWeight", dict(A=25.0, gamma=4.5, s=0.5)),
        ("PotentialPairConservativeGeneralWeight", dict(A=2.0, gamma=4.5, s=2.0)),
        ("AnisoPotentialPairTwoPatchMorse", dict(M_d=1.8341, M_r=0.0302, r_eq=1.0043, omega=5.0, alpha=0.40, repulsion=False)),
    ]
    for name, d in cases:
        c = getattr(m, name)(["A"])
        c.setParams("A", "A", d)
        back = c.getParams("A", "A")
        assert back.keys() == d.keys()
        for k in d:
            assert back[k] == pytest.approx(d[k], rel=1e-15) and type(back[k]) is type(d[k])
    # only the modes the reference accepts (src/pair.py:215: DPD "none"; aniso pairs: none / shift)
    with pytest.raises(RuntimeError):
        m.PotentialPairDPDThermoGeneralWeightGPU(["A"]).mode = "shift"
    a = m.AnisoPotentialPairTwoPatchMorseGPU(["A"])
    a.mode = "shift"
    with pytest.raises(RuntimeError):
        a.mode = "xplor"
    q = m.PotentialBondQuarticGPU(["A-A", "B-B"])
    dq = dict(k=1434.3, r_0=1.5, b_1=-0.7589, b_2=0.0, U_0=67.2234, sigma=1.0, epsilon=1.0)
    q.setParams("B-B", dq)  # delta defaults to 0 (src/bond.py:153)
    assert q.getParams("B-B") == dict(dq, delta=0.0)
    assert np.array_equal(np.frombuffer(q.params_bytes(), dtype=np.float64).reshape(2, 8)[1], oracle.pack_bond_params("Quartic", dict(dq, delta=0.0)))
    w = m.PotentialBondDoubleWellGPU(["A-A"])
    dw = dict(r_0=1.0, r_1=2.0, U_1=1.0, U_tilt=0.5)
    w.setParams("A-A", dw)
    assert w.getParams("A-A") == dw


def test_hilbert_restatement_is_a_hilbert_curve():
    """The Python restatement of Skilling's axes -> Hilbert index that the GPU test compares
    azp_sorter_keys against: a bijection on the 2^b cube whose consecutive indices are face neighbors."""
    import itertools

    from test_gpu_sorter import _hilbert

    for b in (1, 2, 3, 4):
        n = 1 << b
        cells = list(itertools.product(range(n), repeat=3))
        keys = [_hilbert(x, y, z, b) for x, y, z in cells]
        assert sorted(keys) == list(range(n ** 3))
        order = [c for _, c in sorted(zip(keys, cells))]
        assert all(sum(abs(p - q) for p, q in zip(order[k], order[k + 1])) == 1 for k in range(len(order) - 1))
