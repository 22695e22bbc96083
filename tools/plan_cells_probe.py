"""Time of azp_pair_plan_build_from_cells (cell binning excluded) on the north-star system,
whole and phase by phase: AZP_PLAN_CELLS_STOP=1|2|4 leaves the kernel after that phase, in the profiling build of
the library only --

    make -C azplugins_amd/csrc variant SRC=pair_plan_cells NAME=pcprof DEFS=-DAZP_PLAN_CELLS_PROFILE
    AZP_LIB_PATH=tools/libazp_pcprof.so AZP_PLAN_CELLS_STOP=2 python tools/plan_cells_probe.py

    python tools/plan_cells_probe.py [--ncell 64] [--reps 10] [--melt 0]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import azplugins_amd as azp
from azplugins_amd import synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--ncell", type=int, default=64)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--melt", type=int, default=0, help="NVE steps (kT = 1) + a particle sort before the probe")
args = ap.parse_args()

cfg = syn.config_north_star(args.ncell)
sim = azp.Simulation(device="cuda:0", seed=1)
sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"]))
nl = azp.nlist.Cell(buffer=cfg["r_buff"])
pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=cfg["r_cut"])
pot.params[("A", "A")] = cfg["params"]
sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot], methods=[azp.ConstantVolume()])
sim.operations.tuners.clear()
stop = int(os.environ.pop("AZP_PLAN_CELLS_STOP", "0"))   # (handed to the library only for the timed builds)
if args.melt:
    sim.run(0)
    sim.thermalize_particle_momenta(1.0, seed=7)
    sim.run(args.melt)
    azp.ParticleSorter().sort(sim)
sim.run(0)
nl._build(sim.state)         # bins only (fused mode fills no list)
if stop:
    os.environ["AZP_PLAN_CELLS_STOP"] = str(stop)
a = pot._pair_args(for_launch=True)   # (keeps the list fused: no HOOMD-format rows, the bins stay as they are)
stream = torch.cuda.current_stream().cuda_stream
plan = azp._lib.PairPlan()
cells = nl.cells_args(160)
plan.build_from_cells(cells, a, stream)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.reps):
    plan.build_from_cells(cells, a, stream)
e1.record()
torch.cuda.synchronize()
print("stop_after=%d  cell_subdivision=%d  build_from_cells %.3f ms  info %s" % (stop, cells.cell_subdivision, e0.elapsed_time(e1) / args.reps, plan.info()))
