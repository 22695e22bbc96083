"""The isotropic evaluators other than PerturbedLJ on the north-star geometry (N = 2^20 jittered FCC, rho* = 0.8,
r_cut = 3.0, buffer 0.4, <n> = 136.3): tile kernel (plan compiled from the cells, static list, whole in-range rows:
displacement bound 0) and generic kernel, HIP events around --reps launches. Rows for DESIGN / profiles.

    python tools/evaluator_probe.py [hertz yukawa colloid_ss colloid_cc colloid_mix dpd_cons plj] [--reps 30]

colloid_mix: two particle types, half colloids (a = 0.3) half solvent (a = 0): colloid-colloid, colloid-solvent and
solvent-solvent pairs in one launch (the per-type-pair table path of the kernel).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import azplugins_amd as azp
from azplugins_amd import synthetic as syn

ALL = ["plj", "hertz", "yukawa", "dpd_cons", "colloid_ss", "colloid_cc", "colloid_mix"]
ap = argparse.ArgumentParser()
ap.add_argument("which", nargs="*", default=ALL)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--side", type=int, default=64)
ap.add_argument("--settle-ms", type=float, default=80.0)
args = ap.parse_args()

cfg = syn.config_north_star(args.side)
N = cfg["xyz"].shape[0]
for which in (args.which or ALL):
    two = which == "colloid_mix"
    types = ("C", "S") if two else ("A",)
    typeid = (np.arange(N) % 2) if two else None
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], typeid=typeid, types=types))
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    if which == "plj":
        pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=3.0)
        pot.params[("A", "A")] = cfg["params"]
    elif which == "hertz":
        pot = azp.pair.Hertz(nlist=nl, default_r_cut=3.0)
        pot.params[("A", "A")] = dict(epsilon=2.0)
    elif which == "yukawa":
        pot = azp.pair.ExpandedYukawa(nlist=nl, default_r_cut=3.0)
        pot.params[("A", "A")] = dict(epsilon=1.0, kappa=1.2, delta=0.1)
    elif which == "dpd_cons":
        pot = azp.pair.DPDConservativeGeneralWeight(nlist=nl, default_r_cut=3.0) if hasattr(azp.pair, "DPDConservativeGeneralWeight") else None
        if pot is None:
            continue
        pot.params[("A", "A")] = dict(A=25.0, gamma=4.5, s=0.5)
    elif which == "colloid_ss":
        pot = azp.pair.Colloid(nlist=nl, default_r_cut=3.0)
        pot.params[("A", "A")] = dict(A=40.0, a_1=0.0, a_2=0.0, sigma=0.5)
    elif which == "colloid_cc":
        pot = azp.pair.Colloid(nlist=nl, default_r_cut=3.0)
        pot.params[("A", "A")] = dict(A=40.0, a_1=0.3, a_2=0.3, sigma=0.5)
    elif which == "colloid_mix":
        pot = azp.pair.Colloid(nlist=nl, default_r_cut=3.0)
        pot.params[("C", "C")] = dict(A=40.0, a_1=0.3, a_2=0.3, sigma=0.5)
        pot.params[("C", "S")] = dict(A=40.0, a_1=0.3, a_2=0.0, sigma=0.5)
        pot.params[("S", "S")] = dict(A=40.0, a_1=0.0, a_2=0.0, sigma=0.5)
    else:
        raise SystemExit("unknown evaluator %r" % which)
    sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot])
    sim.run(0)

    def settle(ms=args.settle_ms):
        # the chip's power controller cuts the clock 1-4 ms into a burst and recovers over ~50 ms
        # (profiles/r03_clock_transient.md): time at the sustained clock, as bench.py does
        import time

        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(20):
                pot.compute(0)
            torch.cuda.synchronize()

    def timed():
        settle()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            pot.compute(0)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.reps

    mean_n = nl.n_pairs / N
    b_alg = (76.0 + 4.0 * mean_n) * N
    t_plan = timed()
    f_plan = pot.force_tensor.clone()
    pot.use_displacement_bound = False
    t_whole = timed()
    pot.use_plan = False
    t_gen = timed()
    err = float((pot.force_tensor - f_plan).abs().max() / pot.force_tensor.abs().max())
    print("%-12s N=%d <n>=%.1f  tile kernel %.4f ms (bound 0: %.3f of 8 TB/s), whole rows %.4f ms (%.3f); generic %.4f ms; "
          "tile vs generic max|df|/max|f| = %.1e; finite %s" % (which, N, mean_n, t_plan, b_alg / (t_plan * 1e-3) / 8e12, t_whole,
                                                                  b_alg / (t_whole * 1e-3) / 8e12, t_gen, err, bool(torch.isfinite(f_plan).all())))
    del sim, nl, pot
    torch.cuda.empty_cache()
