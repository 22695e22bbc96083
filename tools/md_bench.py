"""End-to-end MD step rate on the north-star system: velocity-Verlet NVE with the
PerturbedLJ pair force, neighbor-list rebuilds on HOOMD's displacement criterion
and tile-plan rebuilds included -- what bench.py's static-list metric leaves out.

    python tools/md_bench.py [--steps 400] [--kT 1.0] [--dt 0.005] [--no-plan]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import azplugins_amd as azp
from azplugins_amd import synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--kT", type=float, default=1.0)
ap.add_argument("--dt", type=float, default=None, help="default 0.005 (c3: 0.002, c5: 0.004)")
ap.add_argument("--ncell", type=int, default=64)
ap.add_argument("--no-plan", action="store_true")
ap.add_argument("--sort-period", type=int, default=200, help="re-index the particles every this many steps (0 = never)")
ap.add_argument("--curve", default="hilbert", help="particle sorter: hilbert | blocks")
ap.add_argument("--tune-buffer", action="store_true", help="let azplugins_amd.tune.NeighborListBuffer pick the neighbor-list buffer first")
ap.add_argument("--workload", default="ns", help="ns: the north-star liquid (PerturbedLJ); c3: BASELINE configs[2], 32,768 chains of 32 beads, "
                                                 "PerturbedLJ + DoubleWell bonds")
ap.add_argument("--buffer", type=float, default=None, help="neighbor-list buffer r_buff (default: the workload's 0.4; HOOMD users tune it)")
args = ap.parse_args()
if args.dt is None:
    args.dt = {"c3": 0.002, "c5": 0.004}.get(args.workload, 0.005)

cfg = {"ns": lambda: syn.config_north_star(args.ncell), "c3": syn.config_chains, "c5": syn.config_tpm, "c4": syn.config_dpd}[args.workload]()
N = cfg["xyz"].shape[0]
sim = azp.Simulation(device="cuda:0", seed=1)
snap = azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], bonds=cfg.get("bonds") if args.workload == "c3" else None, orientation=cfg.get("orientation"),
                                velocity=cfg.get("vel"), tag=cfg.get("tag"))
if args.workload == "c5":
    snap.particles.moment_inertia[:] = [0.1, 0.12, 0.14]
sim.create_state_from_snapshot(snap)
nl = azp.nlist.Cell(buffer=args.buffer if args.buffer is not None else cfg["r_buff"])
if args.workload == "c5":
    pot = azp.pair.TwoPatchMorse(nlist=nl, default_r_cut=cfg["r_cut"], mode="shift")
elif args.workload == "c4":
    pot = azp.pair.DPDGeneralWeight(nlist=nl, kT=cfg["kT"], default_r_cut=cfg["r_cut"])
else:
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=cfg["r_cut"])
pot.params[("A", "A")] = cfg["params"]
pot.use_plan = not args.no_plan
forces = [pot]
if args.workload == "c3":
    dw = azp.bond.DoubleWell()
    dw.params["A-A"] = cfg["bond_params"]
    forces.append(dw)
sim.operations.integrator = azp.Integrator(dt=args.dt, forces=forces, methods=[azp.ConstantVolume()], integrate_rotational_dof=(args.workload == "c5"))
sim.operations.tuners.clear()
if args.sort_period:
    sim.operations.tuners.append(azp.ParticleSorter(trigger_period=args.sort_period, curve=args.curve))
sim.run(0)
if args.workload != "c4":
    sim.thermalize_particle_momenta(args.kT, seed=7)
sim.run(50)  # melt the lattice a little, warm the allocator
if args.sort_period:
    # the first particle sort of a process loads its kernels (6-25 ms of host time): pay that here, the sorts of the
    # timed run (one per --sort-period steps) then cost what they cost in a long run
    sim.operations.tuners[0].sort(sim)
    sim.operations.tuners[0].host_seconds = 0.0
    sim.operations.tuners[0].num_sorts = 0
    sim.run(10)
# likewise the HOOMD-format list (built here once: the path a refused plan compile falls back to drags in a handful of
# framework kernels, tens of ms of module loading the first time)
_ = nl.size
sim.run(10)
if args.tune_buffer:
    tuner = azp.tune.NeighborListBuffer(nl)
    best = tuner.tune(sim)
    print("NeighborListBuffer: steps/s by buffer %s -> buffer %.2f" % ({k: round(v) for k, v in tuner.results.items()}, best))
torch.cuda.synchronize()
b0 = nl.num_builds
e0 = sum(f.energy for f in forces) + 0.5 * float((sim.state.vel[:N, 3] * (sim.state.vel[:N, :3] ** 2).sum(1)).sum())
t0 = time.perf_counter()
sim.run(args.steps)
torch.cuda.synchronize()
t = time.perf_counter() - t0
e1 = sum(f.energy for f in forces) + 0.5 * float((sim.state.vel[:N, 3] * (sim.state.vel[:N, :3] ** 2).sum(1)).sum())
builds = nl.num_builds - b0
print("N=%d  %d steps in %.3f s: %.3f ms/step, %.3e particle-steps/s; %d neighbor-list (+plan) rebuilds = one per %.1f steps; "
      "kT=%.3f; energy drift %.2e per particle" % (N, args.steps, t, 1e3 * t / args.steps, N * args.steps / t, builds,
                                                   args.steps / max(builds, 1), sim.kinetic_temperature(), (e1 - e0) / N))
# force kernel alone on the final (liquid) state
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
pot.compute(0)
ev0.record()
for _ in range(100):
    pot.compute(0)
ev1.record()
torch.cuda.synchronize()
if args.sort_period:
    print("particle sorts: %d (%.1f ms of host time)" % (sim.operations.tuners[0].num_sorts, 1e3 * getattr(sim.operations.tuners[0], "host_seconds", 0.0)))
print("force kernel on the final state: %.4f ms/launch; mean neighbors %.1f; plan %s" % (
    ev0.elapsed_time(ev1) / 100, nl.n_pairs / N, {k: pot.plan_info[k] for k in ("valid", "lds_slots", "max_stage")} if pot.use_plan else None))
if pot.use_plan and pot._plan is not None:
    import numpy as np

    ts = pot._plan.tile_stage()
    if ts.size:
        q = np.percentile(ts, [0, 25, 50, 75, 90, 99, 100]).astype(int)
        print("staged particles per tile: min %d, quartiles %d / %d / %d, 90 %% %d, 99 %% %d, max %d; tiles <= 1536: %.1f %%, <= 1664: %.1f %%" % (
            q[0], q[1], q[2], q[3], q[4], q[5], q[6], 100.0 * (ts <= 1536).mean(), 100.0 * (ts <= 1664).mean()))
