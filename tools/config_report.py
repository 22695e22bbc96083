#!/usr/bin/env python3
"""Runs every BASELINE.json configuration on ONE MI355X at full size: parity of
the product path (hoomd.azplugins-shaped API -> libazp) against the CPU oracle on
the same snapshot and the same neighbor list, plus kernel timing and algorithmic
bandwidth. Prints a markdown table (kept in profiles/).

    python tools/config_report.py [--quick]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import azplugins_amd as azp  # noqa: E402
import oracle  # noqa: E402
from azplugins_amd import synthetic as syn  # noqa: E402


def timed(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def nlist_host(nl):
    return (nl.n_neigh.cpu().numpy().view(np.uint32), nl.head_list.cpu().numpy().view(np.uint64),
            nl.nlist[: nl.size].cpu().numpy().view(np.uint32))


def relerr(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def run(name, cfg, quick):
    N = cfg["xyz"].shape[0]
    snap = azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], velocity=cfg.get("vel"), orientation=cfg.get("orientation"),
                                    tag=cfg.get("tag"), bonds=cfg.get("bonds"))
    sim = azp.Simulation(device="cuda:0", seed=cfg.get("seed", 1))
    sim.create_state_from_snapshot(snap)
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    pot_name = cfg["potential"]
    if pot_name == "DPDGeneralWeight":
        pot = azp.pair.DPDGeneralWeight(nlist=nl, kT=cfg["kT"], default_r_cut=cfg["r_cut"])
    elif pot_name == "TwoPatchMorse":
        pot = azp.pair.TwoPatchMorse(nlist=nl, default_r_cut=cfg["r_cut"], mode="shift")
    else:
        pot = getattr(azp.pair, pot_name)(nlist=nl, default_r_cut=cfg["r_cut"], mode="shift")
    pot.params[("A", "A")] = cfg["params"]
    forces = [pot]
    bond = None
    if "bonds" in cfg:
        bond = getattr(azp.bond, cfg["bond_potential"])()
        bond.params["A-A"] = cfg["bond_params"]
        forces.append(bond)
    sim.operations.integrator = azp.Integrator(dt=cfg.get("dt", 0.005), forces=forces)
    sim.timestep = 12345
    sim.run(0)
    mean_n = nl.n_pairs / N
    rows = []
    # ---- parity on the same snapshot and the same list
    pos = syn.pos4(cfg["xyz"])
    box = oracle.make_box(cfg["L"])
    onl = nlist_host(nl)
    t0 = time.time()
    if pot_name == "DPDGeneralWeight":
        vel = np.zeros((N, 4)); vel[:, :3] = cfg["vel"]; vel[:, 3] = 1.0
        ref = oracle.dpd_forces(pos, vel, cfg["tag"], box, onl, oracle.pack_pair_params(pot_name, cfg["params"]), cfg["r_cut"],
                                kT=cfg["kT"], dt=cfg["dt"], seed=cfg["seed"], timestep=12345)
        err = relerr(np.c_[pot.forces, pot.energies], ref)
        extra = 36
    elif pot_name == "TwoPatchMorse":
        ref, tref = oracle.aniso_forces_tpm(pos, cfg["orientation"], box, onl, oracle.pack_pair_params(pot_name, cfg["params"]),
                                            cfg["r_cut"], mode="shift")
        err = max(relerr(np.c_[pot.forces, pot.energies], ref), relerr(pot.torques, tref[:, :3]))
        extra = 64
    else:
        ref = oracle.pair_forces(pot_name, pos, box, onl, oracle.pack_pair_params(pot_name, cfg["params"]), cfg["r_cut"],
                                 mode="shift", nthreads=min(len(os.sched_getaffinity(0)), 16))
        err = relerr(np.c_[pot.forces, pot.energies], ref)
        extra = 0
    t_oracle = time.time() - t0
    ms = timed(lambda: pot.compute(12345), reps=20 if quick else 100)
    b_alg = 76 + 4 * mean_n + extra
    kern = "tiled (plan)" if (pot.plan_info or {}).get("valid") else "generic"
    rows.append((name, pot_name, N, mean_n, kern, ms, N / ms * 1e-6, b_alg * N / ms * 1e-6, err, t_oracle))
    if bond is not None:
        bref, bad = oracle.bond_forces(cfg["bond_potential"], pos, box, cfg["bonds"], np.zeros(len(cfg["bonds"]), dtype=np.uint32),
                                       oracle.pack_bond_params(cfg["bond_potential"], cfg["bond_params"]))
        berr = relerr(np.c_[bond.forces, bond.energies], bref)
        bms = timed(lambda: bond.compute(0), reps=20 if quick else 100)
        nb = 2.0 * len(cfg["bonds"]) / N
        rows.append((name, cfg["bond_potential"] + " (bond)", N, nb, "bond", bms, N / bms * 1e-6, (68 + 12 * nb) * N / bms * 1e-6, berr, 0.0))
    del sim, pot, nl
    torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated config names (e.g. C4,C5)")
    args = ap.parse_args()
    cfgs = [("C1", syn.config_c1()), ("C2", syn.config_plj_sc(64)), ("NS", syn.config_north_star(64)),
            ("C3", syn.config_chains()), ("C4", syn.config_dpd()), ("C5", syn.config_tpm())]
    if args.quick:
        cfgs = [("C1", syn.config_c1()), ("C2-24", syn.config_plj_sc(24)), ("C3-small", syn.config_chains(32, 16, 16, 16)),
                ("C4-small", syn.config_dpd(32768)), ("C5-small", syn.config_tpm(16, 16, 16))]
    rows = []
    if args.only:
        cfgs = [c for c in cfgs if c[0] in args.only.split(",")]
    for name, cfg in cfgs:
        rows += run(name, cfg, args.quick)
        print("done", name, file=sys.stderr, flush=True)
    print("| config | potential | N | <n> | kernel | ms/launch | 1e9 particle-steps/s | algorithmic GB/s | max err vs oracle (rel. to max) | oracle s |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %s | %d | %.1f | %s | %.4f | %.3f | %.0f | %.1e | %.1f |" % r)


if __name__ == "__main__":
    main()
