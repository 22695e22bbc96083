"""Regenerates the committed golden fixtures from the CPU oracle.

    python tests/golden/make_golden.py

* c1_hertz_forces.npz -- BASELINE.json configs[0] in full: Hertz, N=4,096
  uniform random soft spheres (seed 1), cubic L=16, epsilon=1, r_cut=1.0,
  buffer 0.4, mode none; forces/energies (N,4) from the oracle's HOOMD-style
  CPU loop (half list + third-law scatter).
* sweeps.npz -- per-evaluator dense r-sweeps (256 r values x parameter sets x
  {shift off, on}) of (force_divr, pair_eng) from the oracle's scalar
  evaluators, including the branch edges (r ~ r_wca, r_cut, r_eq, r_0).

* c3_sample.npz, c4_sample.npz, c5_sample.npz -- BASELINE.json configs[2..4] at full
  size (N = 1,048,576 / 2,097,152 / 524,288): forces (and torques) of every
  (N / 1024)-th particle plus sums over all particles, from the oracle
  (`python tests/golden/make_golden.py full`, a few minutes of CPU).

The oracle itself is pinned by the reference's known-answer cases
(reference_cases.json); these fixtures freeze its output so that a later change
to the oracle cannot silently move the target.
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle  # noqa: E402
from azplugins_amd import synthetic as syn  # noqa: E402


def c1():
    cfg = syn.config_c1()
    pos = syn.pos4(cfg["xyz"])
    box = oracle.make_box(cfg["L"])
    nl = oracle.build_nlist(pos, box, cfg["r_cut"] + cfg["r_buff"], half=True)
    params = oracle.pack_pair_params("Hertz", cfg["params"])
    f = oracle.pair_forces("Hertz", pos, box, nl, params, cfg["r_cut"], half=True)
    np.savez_compressed(os.path.join(HERE, "c1_hertz_forces.npz"), force=f, mean_neighbors=2.0 * nl[0].mean())


SWEEPS = {
    "PerturbedLennardJones": (3.0, [dict(epsilon=1.0, sigma=1.0, attraction_scale_factor=0.5),
                                    dict(epsilon=2.0, sigma=1.05, attraction_scale_factor=0.0),
                                    dict(epsilon=0.7, sigma=2.9, attraction_scale_factor=1.0)]),  # rcut < rwca
    "Hertz": (1.5, [dict(epsilon=2.0), dict(epsilon=0.5), dict(epsilon=0.0)]),
    "ExpandedYukawa": (3.0, [dict(epsilon=1.0, kappa=1.0, delta=0.5), dict(epsilon=3.0, kappa=3.0, delta=0.0),
                             dict(epsilon=1.0, kappa=0.1, delta=0.2)]),
    "Colloid": (6.0, [dict(A=100.0, a_1=0.0, a_2=0.0, sigma=2.0), dict(A=100.0, a_1=1.5, a_2=0.0, sigma=1.05),
                      dict(A=100.0, a_1=1.5, a_2=0.75, sigma=1.05)]),
    "DPDConservative": (1.0, [dict(A=25.0, gamma=4.5, s=0.5), dict(A=2.0, gamma=1.0, s=2.0), dict(A=0.0, gamma=4.5, s=2.0)]),
}
SWEEP_RMIN = {"PerturbedLennardJones": 0.8, "Hertz": 0.05, "ExpandedYukawa": 0.55, "Colloid": 2.6, "DPDConservative": 0.05}


def sweeps():
    out = {}
    for name, (r_cut, plist) in SWEEPS.items():
        r = np.linspace(SWEEP_RMIN[name], 1.05 * r_cut, 256)
        # add exact branch edges
        if name == "PerturbedLennardJones":
            r[100] = 2.0 ** (1.0 / 6.0)
        r[200] = r_cut
        res = np.zeros((len(plist), 2, 256, 3))
        for ip, p in enumerate(plist):
            for ish, sh in enumerate((False, True)):
                for ir, rr in enumerate(r):
                    ok, f, e = oracle.eval_pair(name, p, rr, r_cut, sh)
                    res[ip, ish, ir] = (ok, f, e)
        out[name + "_r"] = r
        out[name] = res
    r = np.linspace(0.6, 2.6, 256)
    dw = [dict(r_0=1.0, r_1=2.0, U_1=1.0, U_tilt=0.5), dict(r_0=0.5, r_1=2.5, U_1=5.0, U_tilt=0.0)]
    qt = [dict(k=1434.3, r_0=1.5, b_1=-0.7589, b_2=0.0, U_0=67.2234, sigma=1.0, epsilon=1.0, delta=0.0),
          dict(k=1434.3, r_0=1.5, b_1=-0.7589, b_2=0.0, U_0=67.2234, sigma=1.0, epsilon=1.0, delta=0.5)]
    for name, plist in (("DoubleWell", dw), ("Quartic", qt)):
        res = np.zeros((len(plist), 256, 3))
        for ip, p in enumerate(plist):
            for ir, rr in enumerate(r):
                res[ip, ir] = oracle.eval_bond(name, p, rr)
        out[name + "_r"] = r
        out[name] = res
    np.savez_compressed(os.path.join(HERE, "sweeps.npz"), **out)


N_SAMPLE = 1024


def _sample_and_sums(n, *arrays):
    """Every (n / 1024)-th particle in full precision, plus sums over ALL particles: the
    plain sum and the sum of absolute values of every column (SURVEY 8c golden item 4)."""
    idx = (np.arange(N_SAMPLE, dtype=np.int64) * n) // N_SAMPLE
    out = dict(sample_index=idx)
    for k, a in enumerate(arrays):
        out["sample_%d" % k] = a[idx]
        out["sum_%d" % k] = a.sum(axis=0)
        out["abssum_%d" % k] = np.abs(a).sum(axis=0)
    return out


def c3_c4_c5():
    """BASELINE.json configs[2..4] at full size from the oracle (HOOMD-style loops, all
    cores): C3 = PerturbedLJ (mode shift) with bonded exclusions + DoubleWell bonds,
    N = 1,048,576; C4 = DPD thermostat N = 2,097,152 (seed 7, timestep 0); C5 =
    TwoPatchMorse N = 524,288 (mode shift): 1,024-particle samples + checksums."""
    nt = os.cpu_count() or 8
    # ---- C3
    cfg = syn.config_chains()
    pos = syn.pos4(cfg["xyz"])
    n = pos.shape[0]
    box = oracle.make_box(cfg["L"])
    n_excl = np.zeros(n, dtype=np.uint32)
    excl = np.zeros((n, 2), dtype=np.uint32)
    b = cfg["bonds"].astype(np.int64)
    for me, other in ((b[:, 0], b[:, 1]), (b[:, 1], b[:, 0])):
        order = np.argsort(me, kind="stable")
        m, o = me[order], other[order]
        first = np.r_[True, m[1:] != m[:-1]]
        rank = np.arange(m.size) - np.maximum.accumulate(np.where(first, np.arange(m.size), 0))
        excl[m, n_excl[m] + rank] = o
        np.add.at(n_excl, m, 1)
    nl = oracle.build_nlist(pos, box, cfg["r_cut"] + cfg["r_buff"], exclusions=(n_excl, excl))
    f_pair = oracle.pair_forces("PerturbedLennardJones", pos, box, nl, oracle.pack_pair_params("PerturbedLennardJones", cfg["params"]),
                                cfg["r_cut"], mode="shift", nthreads=nt)
    f_bond, bad = oracle.bond_forces("DoubleWell", pos, box, cfg["bonds"], np.zeros(len(cfg["bonds"]), dtype=np.uint32),
                                     oracle.pack_bond_params("DoubleWell", cfg["bond_params"]))
    assert bad == 0
    np.savez_compressed(os.path.join(HERE, "c3_sample.npz"), mean_neighbors=nl[0].mean(), **_sample_and_sums(n, f_pair, f_bond))
    del nl
    # ---- C4
    cfg = syn.config_dpd()
    pos = syn.pos4(cfg["xyz"])
    n = pos.shape[0]
    vel = np.zeros((n, 4))
    vel[:, :3] = cfg["vel"]
    vel[:, 3] = 1.0
    box = oracle.make_box(cfg["L"])
    nl = oracle.build_nlist(pos, box, cfg["r_cut"] + cfg["r_buff"], half=True)
    f = oracle.dpd_forces(pos, vel, cfg["tag"], box, nl, oracle.pack_pair_params("DPDGeneralWeight", cfg["params"]), cfg["r_cut"],
                          cfg["kT"], cfg["dt"], cfg["seed"], 0, half=True)
    np.savez_compressed(os.path.join(HERE, "c4_sample.npz"), mean_neighbors=2.0 * nl[0].mean(), **_sample_and_sums(n, f))
    del nl
    # ---- C5
    cfg = syn.config_tpm()
    pos = syn.pos4(cfg["xyz"])
    n = pos.shape[0]
    box = oracle.make_box(cfg["L"])
    nl = oracle.build_nlist(pos, box, cfg["r_cut"] + cfg["r_buff"], half=True)
    f, t = oracle.aniso_forces_tpm(pos, cfg["orientation"], box, nl, oracle.pack_pair_params("TwoPatchMorse", cfg["params"]),
                                   cfg["r_cut"], mode="shift", half=True)
    np.savez_compressed(os.path.join(HERE, "c5_sample.npz"), mean_neighbors=2.0 * nl[0].mean(), **_sample_and_sums(n, f, t))


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "sweeps", "full"]
    if "c1" in which:
        c1()
    if "sweeps" in which:
        sweeps()
    if "full" in which:
        c3_c4_c5()
    print("golden fixtures written to", HERE)
