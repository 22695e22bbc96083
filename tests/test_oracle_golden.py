"""Pins the CPU oracle against every known-answer vector the reference's own
tests hold for the hot path (tests/golden/reference_cases.json), through the
same whole-stack observables the reference asserts: per-particle energies
[e/2, e/2] and forces [[-f,0,0],[f,0,0]] (src/pytest/test_pair.py:351-363),
with the reference's tolerance (4 decimals) -- plus tighter checks where the
reference gives more digits."""

import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
with open(os.path.join(GOLDEN, "reference_cases.json")) as _f:
    CASES = json.load(_f)


def _two_particles(d):
    return np.array([[-d / 2, 0.0, 0.0], [d / 2, 0.0, 0.0]])


def _ids(cases):
    return ["%s@%s" % (c["potential"], c["src"].split(":")[-1]) for c in cases]


@pytest.mark.parametrize("case", CASES["pair"], ids=_ids(CASES["pair"]))
@pytest.mark.parametrize("half", [True, False], ids=["half", "full"])
def test_pair_known_answers(oracle, case, half):
    r_cut = case["r_cut"]
    L = 2.1 * 2 * (r_cut + 0.4)  # src/pytest/test_pair.py:318-322
    pos = oracle.pos4(_two_particles(case["distance"]))
    box = oracle.make_box(L)
    nl = oracle.build_nlist(pos, box, r_cut + 0.4, half=half)
    name = case["potential"]
    if name == "DPDGeneralWeight":
        params = oracle.pack_pair_params(name, case["params"])
        vel = np.zeros((2, 4))
        vel[:, 3] = 1.0
        force = oracle.dpd_forces(pos, vel, np.arange(2), box, nl, params, r_cut, kT=case["kT"], dt=0.001, seed=1,
                                  timestep=0, half=half)
    else:
        params = oracle.pack_pair_params(name, case["params"])
        force = oracle.pair_forces(name, pos, box, nl, params, r_cut, mode="shift" if case["shift"] else "none",
                                   half=half)
    e, f = case["energy"], case["force"]
    np.testing.assert_array_almost_equal(force[:, 3], [0.5 * e, 0.5 * e], decimal=4)
    np.testing.assert_array_almost_equal(force[:, :3], [[-f, 0, 0], [f, 0, 0]], decimal=4)


def test_pair_many_digit_cases(oracle):
    """ExpandedYukawa cases carry 16 significant digits in the reference."""
    n = 0
    for case in CASES["pair"]:
        if case["potential"] != "ExpandedYukawa" or case["energy"] == 0:
            continue
        ok, fdivr, e = oracle.eval_pair("ExpandedYukawa", case["params"], case["distance"], case["r_cut"], case["shift"])
        assert ok
        assert e == pytest.approx(case["energy"], rel=1e-13)
        assert fdivr * case["distance"] == pytest.approx(case["force"], rel=2e-10)
        n += 1
    assert n == 5


@pytest.mark.parametrize("case", CASES["aniso"], ids=_ids(CASES["aniso"]))
@pytest.mark.parametrize("half", [True, False], ids=["half", "full"])
def test_aniso_known_answers(oracle, case, half):
    setup = CASES["aniso_setup"]
    pos = oracle.pos4(np.array(setup["positions"]))
    q = np.array(setup["orientations"], dtype=np.float64)
    box = oracle.make_box(20.0)  # two_particle_snapshot_factory default L=20
    r_cut = case["r_cut"]
    nl = oracle.build_nlist(pos, box, r_cut + 0.4, half=half)
    params = oracle.pack_pair_params("TwoPatchMorse", case["params"])
    force, torque = oracle.aniso_forces_tpm(pos, q, box, nl, params, r_cut, mode="shift" if case["shift"] else "none",
                                            half=half)
    e = case["energy"]
    np.testing.assert_array_almost_equal(force[:, 3], [0.5 * e, 0.5 * e], decimal=4)
    if case["force"] is not None:
        f = np.array(case["force"])
        np.testing.assert_array_almost_equal(force[:, :3], [-f, f], decimal=4)
    if case["torque"] is not None:
        T = np.array(case["torque"])
        np.testing.assert_array_almost_equal(torque[:, :3], [T, T], decimal=4)


@pytest.mark.parametrize("case", CASES["bond"], ids=_ids(CASES["bond"]))
def test_bond_known_answers(oracle, case):
    pos = oracle.pos4(_two_particles(case["distance"]))
    box = oracle.make_box(20.0)
    params = oracle.pack_bond_params(case["potential"], case["params"])
    force, bad = oracle.bond_forces(case["potential"], pos, box, [[0, 1]], [0], params)
    assert bad == 0
    e, f = case["energy"], case["force"]
    np.testing.assert_array_almost_equal(force[:, 3], [0.5 * e, 0.5 * e], decimal=4)
    np.testing.assert_array_almost_equal(force[:, :3], [[-f, 0, 0], [f, 0, 0]], decimal=4)
    # these cases are given to >= 8 digits by the reference
    ok, fdivr, eng = oracle.eval_bond(case["potential"], case["params"], case["distance"])
    assert ok
    assert eng == pytest.approx(e, rel=1e-8, abs=1e-9)
    assert fdivr * case["distance"] == pytest.approx(f, rel=1e-8, abs=1e-9)


def test_bond_invalid_params_flagged(oracle):
    """r_diff == 0 (src/BondEvaluatorDoubleWell.h:101-102) and r_0 == 0
    (src/BondEvaluatorQuartic.h:134-135) make the evaluator return false."""
    pos = oracle.pos4(_two_particles(1.0))
    box = oracle.make_box(20.0)
    p = oracle.pack_bond_params("DoubleWell", dict(r_0=1.0, r_1=1.0, U_1=1.0, U_tilt=0.0))
    force, bad = oracle.bond_forces("DoubleWell", pos, box, [[0, 1]], [0], p)
    assert bad == 1 and not force.any()
    p = oracle.pack_bond_params("Quartic", dict(k=1.0, r_0=0.0, b_1=0, b_2=0, U_0=0, sigma=1, epsilon=1, delta=0))
    force, bad = oracle.bond_forces("Quartic", pos, box, [[0, 1]], [0], p)
    assert bad == 1 and not force.any()


def test_philox_known_answers(oracle):
    for c in CASES["philox4x32_10_kat"]["cases"]:
        ctr = [int(x, 16) for x in c["ctr"]]
        key = [int(x, 16) for x in c["key"]]
        out = oracle.philox4x32_10(ctr, key)
        assert [int(x) for x in out] == [int(x, 16) for x in c["out"]]


def test_dpd_thermo_closed_form(oracle):
    """Drag + random terms of src/DPDPairEvaluatorGeneralWeight.h:236-246 in
    closed form with an injected alpha."""
    A, gamma, s, rc, r, dt, kT, alpha, rdotv = 2.0, 4.5, 0.5, 1.0, 0.5, 0.01, 1.5, 0.37, -0.21
    ok, f, fc, e = oracle.eval_dpd_thermo(dict(A=A, gamma=gamma, s=s), r, rc, rdotv, dt, kT, alpha)
    assert ok
    wR = (1 - r / rc) ** (0.5 * s) / r
    assert fc == pytest.approx(A * (1 / r - 1 / rc), rel=1e-14)
    expect = fc - gamma * wR * wR * rdotv + np.sqrt(6 * kT * gamma / dt) * wR * alpha
    assert f == pytest.approx(expect, rel=1e-13)
    assert e == pytest.approx(A * (rc - r) - 0.5 * A / rc * (rc * rc - r * r), rel=1e-14)
    # kT = 0 => no noise
    ok, f0, fc0, _ = oracle.eval_dpd_thermo(dict(A=A, gamma=gamma, s=s), r, rc, 0.0, dt, 0.0, alpha)
    assert f0 == fc0


def test_dpd_alpha_symmetric_and_uniform(oracle):
    a = np.array([oracle.dpd_alpha(7, i, i + 1 + (i % 5), 3) for i in range(20000)])
    assert oracle.dpd_alpha(7, 11, 5, 3) == oracle.dpd_alpha(7, 5, 11, 3)
    assert oracle.dpd_alpha(7, 11, 5, 3) != oracle.dpd_alpha(7, 11, 5, 4)
    assert oracle.dpd_alpha(8, 11, 5, 3) != oracle.dpd_alpha(7, 11, 5, 3)
    assert -1.0 < a.min() and a.max() <= 1.0
    assert abs(a.mean()) < 0.02
    assert a.var() == pytest.approx(1.0 / 3.0, rel=0.03)


def test_oracle_reproduces_committed_sweeps(oracle):
    """tests/golden/sweeps.npz freezes the oracle's scalar evaluators over dense
    r-sweeps (incl. the branch edges r = r_wca, r = r_cut): a later edit of the
    oracle cannot silently move the target."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(os.path.join(GOLDEN, "sweeps.npz"))
    for name, (r_cut, plist) in mg.SWEEPS.items():
        r = gold[name + "_r"]
        ref = gold[name]
        for ip, p in enumerate(plist):
            for ish, sh in enumerate((False, True)):
                for ir in range(0, len(r), 7):
                    ok, f, e = oracle.eval_pair(name, p, float(r[ir]), r_cut, sh)
                    assert (float(ok), f, e) == tuple(ref[ip, ish, ir])


@pytest.mark.parametrize("kind", ["spherical", "planar"])
def test_external_barrier_known_answers(oracle, kind):
    """src/pytest/test_external.py:95-223 against the oracle's barrier loop."""
    ext = CASES["external"]
    c = ext[kind]
    pos = oracle.pos4(np.array(c["positions"], dtype=float), ext["typeid"])
    box = oracle.make_box(ext["box"])
    for run in ("run1", "run2"):
        r = c[run]
        params = [[ext["kA"], ext["offset_A"]], [r["kB"], ext["offset_B"]]]
        f = oracle.barrier_forces(kind, pos, box, params, r["location"])
        np.testing.assert_allclose(f[:, 3], r["energies"], atol=1e-4)
        np.testing.assert_allclose(f[:, :3], r["forces"], atol=1e-4)


def test_rotational_nve_free_top_conserves_momentum_and_energy(oracle):
    """azo_nve_rot_step (the rotational half of the NVE step, PARITY UNPINNED against HOOMD:
    its source is absent) on torque-free asymmetric tops: the space-frame angular momentum
    and the rotational kinetic energy are constants of the motion; the NO_SQUISH free
    rotations conserve |L| to rounding and the energy to O(dt^2); q stays normalised; bodies
    with a zero moment of inertia (no rotation about that axis, no energy term) are included."""
    rng = np.random.default_rng(3)
    n = 64
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1)[:, None]
    I = np.tile(np.array([0.4, 0.7, 1.1]), (n, 1))
    I[:8, 2] = 0.0  # rods: no rotation about the body z axis
    s_body = rng.normal(size=(n, 3))
    s_body[:8, 2] = 0.0

    def qmul(a, b):
        s = a[:, :1] * b[:, :1] - (a[:, 1:] * b[:, 1:]).sum(1, keepdims=True)
        v = a[:, :1] * b[:, 1:] + b[:, :1] * a[:, 1:] + np.cross(a[:, 1:], b[:, 1:])
        return np.concatenate([s, v], axis=1)

    def body(q, p):  # s = 1/2 conj(q) p
        qc = q * np.array([1.0, -1.0, -1.0, -1.0])
        return 0.5 * qmul(qc, p)[:, 1:]

    def space(q, s):  # rotate(q, s)
        qc = q * np.array([1.0, -1.0, -1.0, -1.0])
        return qmul(qmul(q, np.concatenate([np.zeros((len(s), 1)), s], axis=1)), qc)[:, 1:]

    def ke(q, p):
        s = body(q, p)
        return 0.5 * np.where(I != 0.0, s * s / np.where(I != 0.0, I, 1.0), 0.0).sum(axis=1)

    p = 2.0 * qmul(q, np.concatenate([np.zeros((n, 1)), s_body], axis=1))
    assert np.allclose(body(q, p), s_body)
    zero = np.zeros((n, 4))
    L0, E0 = space(q, body(q, p)), ke(q, p)
    drift = []
    for dt in (0.01, 0.005):
        qq, pp = q.copy(), p.copy()
        for _ in range(int(round(2.0 / dt))):
            qq, pp = oracle.nve_rot_step(True, qq, pp, I, zero, dt)
            qq, pp = oracle.nve_rot_step(False, qq, pp, I, zero, dt)
        assert np.allclose(np.linalg.norm(qq, axis=1), 1.0, atol=1e-14)
        assert np.allclose(np.linalg.norm(space(qq, body(qq, pp)), axis=1), np.linalg.norm(L0, axis=1), rtol=1e-10)
        assert np.abs(space(qq, body(qq, pp)) - L0).max() < 2e-3 * (dt / 0.01) ** 2 * np.abs(L0).max() + 1e-12
        drift.append(np.abs(ke(qq, pp) - E0).max() / E0.max())
        assert not np.allclose(qq, q)  # it did rotate
    assert drift[0] < 1e-3 and drift[1] < 0.3 * drift[0] + 1e-12  # second order in dt
    # a constant body-frame torque about a principal axis of a top at rest: s_x grows as torque * t
    q1 = np.tile(np.array([1.0, 0.0, 0.0, 0.0]), (n, 1))
    p1 = np.zeros((n, 4))
    tq = np.zeros((n, 4))
    tq[:, 0] = 0.3
    dt = 0.01
    for _ in range(100):
        q1, p1 = oracle.nve_rot_step(True, q1, p1, I, tq, dt)
        q1, p1 = oracle.nve_rot_step(False, q1, p1, I, tq, dt)
    assert np.allclose(body(q1, p1)[:, 0], 0.3 * 1.0, rtol=1e-10)
