"""The HOOMD-signature entry points (azp_pair_forces_<evaluator>: what the reference's
gpu_compute_pair_forces<E> forwards to, src/PotentialPairGPUKernel.cu.inc:25-28) with
libazp's own plan cache: the caller never says that the neighbor list was rebuilt, moves
particles and rewrites the list in place as HOOMD does -- forces must equal the oracle's at
every step, and the call must run at tile-kernel speed, not at the generic kernel's."""

import ctypes as C
import time

import numpy as np
import pytest

import helpers as H
from azplugins_amd import _lib
from azplugins_amd import synthetic as syn

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _close(got, ref, what=""):
    assert np.all(np.isfinite(got)), what
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= TOL * scale, "%s: %g vs scale %g" % (what, np.abs(got - ref).max(), scale)


def _write_list(t, nl):
    """Rewrite the device list in place (same buffers, HOOMD's NeighborList does the same)."""
    import torch

    n_neigh, head, nlist = nl
    t["n_neigh"].copy_(torch.from_numpy(np.ascontiguousarray(n_neigh, dtype=np.uint32).view(np.int32)))
    t["head"].copy_(torch.from_numpy(np.ascontiguousarray(head, dtype=np.uint64).view(np.int64)))
    t["nlist"][: nlist.size].copy_(torch.from_numpy(np.ascontiguousarray(nlist, dtype=np.uint32).view(np.int32)))


@pytest.mark.parametrize("T", [1, 2])
def test_hoomd_signature_entry_across_list_rebuilds(oracle, T):
    import torch

    r_buff = 0.4
    cfg = syn.config_plj_sc(16)
    n = cfg["xyz"].shape[0]
    L = cfg["L"]
    typeid = (np.arange(n) // 7) % T
    box = oracle.make_box(L)
    r_cut = np.full((T, T), 3.0)
    if T > 1:
        r_cut[0, 1] = r_cut[1, 0] = 2.7
    tab = H.sym_table(T, lambda i, j: dict(epsilon=1.0 + 0.2 * (i + j), sigma=1.0 - 0.02 * (i + j), attraction_scale_factor=0.5))
    params = np.array([oracle.pack_pair_params("PerturbedLennardJones", tab[i][j]) for i in range(T) for j in range(T)])
    pos = syn.pos4(cfg["xyz"], typeid)
    nl = oracle.build_nlist(pos, box, r_cut + r_buff, ntypes=T)
    a, t = H.gpu_pair_args(pos, (L,), nl, T, r_cut, 0.0, "shift", False, auto_plan=True)
    # room for rebuilt lists in the same buffers
    cap = int(nl[2].size * 1.2) + 64
    t["nlist"] = torch.zeros(cap, dtype=torch.int32, device="cuda:0")
    a.d_nlist = t["nlist"].data_ptr()
    _write_list(t, nl)
    p = H._dev(params)
    lib = _lib.lib()
    lib.azp_pair_auto_plan_clear()
    s0 = _lib.auto_plan_stats()

    def call(pos_now, nl_now, what):
        t["pos"].copy_(torch.from_numpy(pos_now))
        t["force"].fill_(float("nan"))
        _lib.check(lib.azp_pair_forces_perturbed_lennard_jones(C.byref(a), p.data_ptr(), H._stream()), what)
        torch.cuda.synchronize()
        f_ref = oracle.pair_forces("PerturbedLennardJones", pos_now, box, nl_now, params, r_cut, 0.0, "shift", ntypes=T, nthreads=8)
        _close(t["force"].cpu().numpy(), f_ref, what)
        return _lib.auto_plan_stats()

    tag = np.arange(n, dtype=np.uint64)

    def moved(frac, seed):
        v = np.stack([syn.normal(seed, tag, c) for c in range(3)], axis=1)
        v *= (frac * 0.5 * r_buff * syn.u01(seed, tag, 5) / np.linalg.norm(v, axis=1))[:, None]
        out = pos.copy()
        out[:, :3] = syn.wrap(pos[:, :3] + v, L)
        return out

    s1 = call(pos, nl, "first call")
    assert s1["compiles"] - s0["compiles"] == 2 and s1["generic_fallbacks"] == s0["generic_fallbacks"]  # learn r_list, then compile
    s2 = call(pos, nl, "unchanged")
    assert s2["compiles"] == s1["compiles"] and s2["reuses"] == s1["reuses"] + 1
    # particles move between two list builds: same list, same plan, displacement measured inside
    for frac in (0.2, 0.6, 0.99):
        s3 = call(moved(frac, 11), nl, "moved %.2f" % frac)
        assert s3["compiles"] == s1["compiles"]
    # HOOMD rebuilds the list in place at the moved positions
    pos_b = moved(0.99, 11)
    nl_b = oracle.build_nlist(pos_b, box, r_cut + r_buff, ntypes=T)
    assert nl_b[2].size <= cap
    _write_list(t, nl_b)
    s4 = call(pos_b, nl_b, "after rebuild")
    assert s4["compiles"] == s1["compiles"] + 1
    # a rebuild that keeps every row length and only permutes entries inside rows
    nl_c = (nl_b[0], nl_b[1], nl_b[2].copy())
    i = int(np.argmax(nl_b[0]))
    h, k = int(nl_b[1][i]), int(nl_b[0][i])
    nl_c[2][h:h + k] = nl_b[2][h:h + k][::-1]
    _write_list(t, nl_c)
    s5 = call(pos_b, nl_c, "row permuted")
    assert s5["compiles"] == s4["compiles"] + 1
    # the cutoff table changes in place (HOOMD: r_cut set by the user; the list on hand still covers it)
    r_cut2 = r_cut - 0.15
    t["rcutsq"].copy_(torch.from_numpy((r_cut2 * r_cut2).reshape(-1)))
    t["force"].fill_(float("nan"))
    _lib.check(lib.azp_pair_forces_perturbed_lennard_jones(C.byref(a), p.data_ptr(), H._stream()), "r_cut changed")
    torch.cuda.synchronize()
    f_ref = oracle.pair_forces("PerturbedLennardJones", pos_b, box, nl_c, params, r_cut2, 0.0, "shift", ntypes=T, nthreads=8)
    _close(t["force"].cpu().numpy(), f_ref, "r_cut changed")
    assert _lib.auto_plan_stats()["compiles"] == s5["compiles"] + 1
    # opting out per call gives the generic kernel and leaves the cache alone
    a.flags = _lib.PAIR_FLAG_NO_AUTO_PLAN
    before = _lib.auto_plan_stats()
    _lib.check(lib.azp_pair_forces_perturbed_lennard_jones(C.byref(a), p.data_ptr(), H._stream()), "generic")
    torch.cuda.synchronize()
    _close(t["force"].cpu().numpy(), f_ref, "generic")
    assert _lib.auto_plan_stats() == before
    lib.azp_pair_auto_plan_clear()


@pytest.mark.parametrize("name", ["Hertz", "ExpandedYukawa", "Colloid", "DPDConservative"])
def test_hoomd_signature_entry_other_evaluators(oracle, name):
    """Every isotropic evaluator behind its HOOMD-signature entry with the plan cache on,
    virial and xplor included, equals the generic kernel's oracle-checked answer."""
    pos, L, typeid = H.lattice_config(12, 1.6 if name == "Colloid" else 1.1, 0.12, seed=8, ntypes=1)
    box = oracle.make_box(L)
    r_cut = 3.2 if name == "Colloid" else 2.5
    d = dict(Hertz=dict(epsilon=2.0), ExpandedYukawa=dict(epsilon=1.0, kappa=1.2, delta=0.1),
             Colloid=dict(A=40.0, a_1=0.3, a_2=0.3, sigma=0.5), DPDConservative=dict(A=25.0, gamma=4.5, s=0.5))[name]
    params = oracle.pack_pair_params(name, d)
    nl = oracle.build_nlist(pos, box, r_cut + 0.3)
    mode = "none" if name == "DPDConservative" else "xplor"
    f_ref, v_ref = oracle.pair_forces(name, pos, box, nl, params, r_cut, 0.8 * r_cut, mode, virial=True)
    _lib.lib().azp_pair_auto_plan_clear()
    s0 = _lib.auto_plan_stats()
    f, v = H.gpu_pair_forces(name, pos, (L,), nl, params, r_cut, 0.8 * r_cut, mode, virial=True, auto_plan=True)
    s1 = _lib.auto_plan_stats()
    assert s1["calls"] == s0["calls"] + 1 and s1["generic_fallbacks"] == s0["generic_fallbacks"]
    _close(f, f_ref, name)
    _close(v, v_ref, name + " virial")
    _lib.lib().azp_pair_auto_plan_clear()


def test_hoomd_signature_entry_runs_at_plan_speed():
    """N = 2^20 (the headline workload): the entry point the adapter calls, timed per call
    on the host (its readback included), against the generic kernel it used to run and the
    explicit plan API. Forces identical to the explicit plan's."""
    import torch

    import azplugins_amd as azp

    cfg = syn.config_north_star(64)
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"]))
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    nl.fused = False            # HOOMD hands its own u32 list to the entry point: build one
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=cfg["r_cut"])
    pot.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot])
    sim.run(0)
    f_plan = pot.force_tensor.clone()
    a = pot._pair_args()
    a.r_list_max = 0.0          # pair_args_t has no such field
    a.has_displacement_bound = 0
    a.d_rinnersq = None
    lib = _lib.lib()
    lib.azp_pair_auto_plan_clear()
    stream = torch.cuda.current_stream().cuda_stream
    fn = lib.azp_pair_forces_perturbed_lennard_jones
    p = pot._tables["params"].data_ptr()

    def timed(reps):
        _lib.check(fn(C.byref(a), p, stream), "entry")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            _lib.check(fn(C.byref(a), p, stream), "entry")
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    a.flags = 0
    ms_auto = timed(20)
    f_auto = pot.force_tensor.clone()
    a.flags = _lib.PAIR_FLAG_NO_AUTO_PLAN
    ms_generic = timed(10)
    f_generic = pot.force_tensor.clone()
    scale = float(f_generic.abs().max())
    assert float((f_auto - f_generic).abs().max()) <= 1e-11 * scale
    assert float((f_auto - f_plan).abs().max()) <= 1e-12 * scale
    st = _lib.auto_plan_stats()
    assert st["reuses"] >= 20 and st["generic_fallbacks"] == 0
    print("HOOMD-signature entry: %.3f ms per call with the plan cache, %.3f ms generic" % (ms_auto, ms_generic))
    assert ms_auto < 0.6 * ms_generic
    lib.azp_pair_auto_plan_clear()


def _moved_positions(pos, L, r_buff, frac, seed):
    n = pos.shape[0]
    tag = np.arange(n, dtype=np.uint64)
    v = np.stack([syn.normal(seed, tag, c) for c in range(3)], axis=1)
    v *= (frac * 0.5 * r_buff * syn.u01(seed, tag, 5) / np.linalg.norm(v, axis=1))[:, None]
    out = pos.copy()
    out[:, :3] = syn.wrap(pos[:, :3] + v, L)
    return out


def test_dpd_thermostat_behind_its_hoomd_signature(oracle):
    """gpu_compute_dpd_forces<E> (src/PotentialPairDPDThermoGPUKernel.cu.inc:21-24) forwards to
    azp_dpd_forces_general_weight: with the plan cache it runs the tile-staged DPD kernel. Particles move,
    HOOMD rewrites the list in place -- forces equal the oracle's at every call; compile / reuse counters as expected."""
    import torch

    r_cut, r_buff = 1.0, 0.4
    cfg = syn.config_dpd(8192)
    n = cfg["xyz"].shape[0]
    L = cfg["L"]
    pos = syn.pos4(cfg["xyz"])
    vel = np.zeros((n, 4))
    vel[:, :3] = cfg["vel"]
    vel[:, 3] = 1.0
    box = oracle.make_box(L)
    params = np.atleast_2d(oracle.pack_pair_params("DPDGeneralWeight", cfg["params"]))
    nl = oracle.build_nlist(pos, box, r_cut + r_buff)
    a, t = H.gpu_pair_args(pos, (L,), nl, 1, r_cut, 0.0, "none", True, auto_plan=True)
    cap = int(nl[2].size * 1.2) + 64
    t["nlist"] = torch.zeros(cap, dtype=torch.int32, device="cuda:0")
    a.d_nlist = t["nlist"].data_ptr()
    _write_list(t, nl)
    p = H._dev(params)
    v_d, tg = H._dev(vel, np.float64), H._dev(cfg["tag"], np.uint32)
    d = _lib.DPDArgs()
    d.d_vel, d.d_tag, d.deltaT, d.T, d.seed = v_d.data_ptr(), tg.data_ptr(), 0.01, 1.0, 7
    lib = _lib.lib()
    lib.azp_pair_auto_plan_clear()
    kw = dict(kT=1.0, dt=0.01, seed=7)

    def call(pos_now, nl_now, step, what):
        t["pos"].copy_(torch.from_numpy(pos_now))
        t["force"].fill_(float("nan"))
        d.pair = a
        d.timestep = step
        _lib.check(lib.azp_dpd_forces_general_weight(C.byref(d), p.data_ptr(), H._stream()), what)
        torch.cuda.synchronize()
        f_ref, v_ref = oracle.dpd_forces(pos_now, vel, cfg["tag"], box, nl_now, params, r_cut, virial=True, timestep=step, **kw)
        _close(t["force"].cpu().numpy(), f_ref, what)
        _close(t["virial"].cpu().numpy(), v_ref, what + " virial")
        return _lib.auto_plan_stats()

    s0 = _lib.auto_plan_stats()
    s1 = call(pos, nl, 100, "first call")
    assert s1["compiles"] - s0["compiles"] == 2 and s1["generic_fallbacks"] == s0["generic_fallbacks"]
    for k, frac in enumerate((0.0, 0.3, 0.7, 0.99)):
        s2 = call(_moved_positions(pos, L, r_buff, frac, 21), nl, 101 + k, "moved %.2f" % frac)
        assert s2["compiles"] == s1["compiles"] and s2["reuses"] == s1["reuses"] + k + 1
    pos_b = _moved_positions(pos, L, r_buff, 0.99, 21)
    nl_b = oracle.build_nlist(pos_b, box, r_cut + r_buff)
    assert nl_b[2].size <= cap
    _write_list(t, nl_b)
    s3 = call(pos_b, nl_b, 200, "after rebuild")  # the speculative launch on the stale plan is overwritten
    assert s3["compiles"] == s2["compiles"] + 1 and s3["generic_fallbacks"] == s0["generic_fallbacks"]
    s4 = call(_moved_positions(pos_b, L, r_buff, 0.5, 22), nl_b, 201, "moved after rebuild")
    assert s4["compiles"] == s3["compiles"]
    lib.azp_pair_auto_plan_clear()


def test_two_patch_morse_behind_its_hoomd_signature(oracle):
    """gpu_compute_pair_aniso_forces<E> (src/AnisoPotentialPairGPUKernel.cu.inc:21-25) forwards to
    azp_aniso_forces_two_patch_morse: tile-staged kernel from the plan cache, forces AND torques equal the oracle's
    while particles move and the list is rewritten in place."""
    import torch

    r_cut, r_buff = 1.6, 0.4
    cfg = syn.config_tpm(12, 12, 12)
    L = cfg["L"]
    pos = syn.pos4(cfg["xyz"])
    n = pos.shape[0]
    box = oracle.make_box(L)
    params = np.atleast_2d(oracle.pack_pair_params("TwoPatchMorse", cfg["params"]))
    nl = oracle.build_nlist(pos, box, r_cut + r_buff)
    a, t = H.gpu_pair_args(pos, (L,), nl, 1, r_cut, 0.0, "shift", True, auto_plan=True)
    cap = int(nl[2].size * 1.3) + 64
    t["nlist"] = torch.zeros(cap, dtype=torch.int32, device="cuda:0")
    a.d_nlist = t["nlist"].data_ptr()
    _write_list(t, nl)
    p = H._dev(params)
    q = H._dev(cfg["orientation"], np.float64)
    tq = torch.full((n, 4), float("nan"), dtype=torch.float64, device="cuda:0")
    g = _lib.AnisoArgs()
    g.d_orientation, g.d_torque = q.data_ptr(), tq.data_ptr()
    lib = _lib.lib()
    lib.azp_pair_auto_plan_clear()

    def call(pos_now, nl_now, what):
        t["pos"].copy_(torch.from_numpy(pos_now))
        t["force"].fill_(float("nan"))
        tq.fill_(float("nan"))
        g.pair = a
        _lib.check(lib.azp_aniso_forces_two_patch_morse(C.byref(g), p.data_ptr(), H._stream()), what)
        torch.cuda.synchronize()
        f_ref, t_ref, v_ref = oracle.aniso_forces_tpm(pos_now, cfg["orientation"], box, nl_now, params, r_cut, "shift", virial=True)
        _close(t["force"].cpu().numpy(), f_ref, what)
        _close(tq.cpu().numpy()[:, :3], t_ref[:, :3], what + " torque")
        _close(t["virial"].cpu().numpy(), v_ref, what + " virial")
        return _lib.auto_plan_stats()

    s0 = _lib.auto_plan_stats()
    s1 = call(pos, nl, "first call")
    assert s1["compiles"] - s0["compiles"] == 2 and s1["generic_fallbacks"] == s0["generic_fallbacks"]
    for k, frac in enumerate((0.4, 0.99)):
        s2 = call(_moved_positions(pos, L, r_buff, frac, 31), nl, "moved %.2f" % frac)
        assert s2["compiles"] == s1["compiles"]
    pos_b = _moved_positions(pos, L, r_buff, 0.99, 31)
    nl_b = oracle.build_nlist(pos_b, box, r_cut + r_buff)
    assert nl_b[2].size <= cap
    _write_list(t, nl_b)
    s3 = call(pos_b, nl_b, "after rebuild")
    assert s3["compiles"] == s2["compiles"] + 1 and s3["generic_fallbacks"] == s0["generic_fallbacks"]
    lib.azp_pair_auto_plan_clear()


def test_sampled_fingerprint_rotates_and_generation_counter(oracle):
    """Lists above 2^22 entries are fingerprinted by row lengths, row starts and two entries of every 8th row, the
    sampled rows rotating from call to call: a rewrite that keeps every length (here: one row reversed) is noticed
    within eight calls. A caller that passes azp_pair_args.list_generation is not fingerprinted at all: the plan is
    recompiled exactly when the number changes."""
    import torch

    r_cut, r_buff = 3.0, 0.4
    cfg = syn.config_plj_sc(14)
    L = cfg["L"]
    pos = syn.pos4(cfg["xyz"])
    box = oracle.make_box(L)
    params = np.atleast_2d(oracle.pack_pair_params("PerturbedLennardJones", cfg["params"]))
    nl = oracle.build_nlist(pos, box, r_cut + r_buff)
    a, t = H.gpu_pair_args(pos, (L,), nl, 1, r_cut, 0.0, "shift", False, auto_plan=True)
    a.size_nlist = (1 << 22) + 1   # claim a long list: the sampled fingerprint
    p = H._dev(params)
    lib = _lib.lib()
    lib.azp_pair_auto_plan_clear()
    fn = lib.azp_pair_forces_perturbed_lennard_jones

    def call(nl_now, what):
        t["force"].fill_(float("nan"))
        _lib.check(fn(C.byref(a), p.data_ptr(), H._stream()), what)
        torch.cuda.synchronize()
        f_ref = oracle.pair_forces("PerturbedLennardJones", pos, box, nl_now, params, r_cut, 0.0, "shift", nthreads=8)
        _close(t["force"].cpu().numpy(), f_ref, what)
        return _lib.auto_plan_stats()

    s1 = call(nl, "first")
    s2 = call(nl, "second")
    assert s2["compiles"] == s1["compiles"]
    # reverse the MIDDLE entries of one row (the sample reads the last and the middle entry of a sampled row: moving
    # the entries changes the middle one). The set of neighbors is the same, so the forces stay right either way;
    # what is asserted is that the change is seen within eight calls.
    i = 5
    h, k = int(nl[1][i]), int(nl[0][i])
    nl_c = (nl[0], nl[1], nl[2].copy())
    nl_c[2][h:h + k - 1] = nl[2][h:h + k - 1][::-1]
    _write_list(t, nl_c)
    seen = None
    for c in range(8):
        s = call(nl_c, "rotating %d" % c)
        if s["compiles"] > s2["compiles"]:
            seen = c
            break
    assert seen is not None
    # generation counter: no fingerprint; recompiled exactly when the number changes
    lib.azp_pair_auto_plan_clear()
    a.list_generation = 41
    g0 = _lib.auto_plan_stats()
    g1 = call(nl_c, "generation 41")
    assert g1["compiles"] > g0["compiles"]
    _write_list(t, nl)       # rewritten in place, same generation: the caller vouches that nothing changed
    g2 = call(nl, "generation 41 again (same neighbor sets)")
    assert g2["compiles"] == g1["compiles"]
    a.list_generation = 42
    g3 = call(nl, "generation 42")
    assert g3["compiles"] == g2["compiles"] + 1
    lib.azp_pair_auto_plan_clear()
