"""Kernel time at every step of one neighbor-list rebuild cycle of a real NVE run.

The north-star lattice gets Maxwell velocities, is advanced with the NVE kernels until
HOOMD's distance check asks for a rebuild, and the positions after every step are kept.
The list and the tile plan are then built once for the first snapshot and the pair
kernel is timed on each snapshot (a) as an MD run would call it -- with the displacement
bound the distance check returns -- and (b) over whole rows.

    python tools/cycle_probe.py [--kT 1.0] [--dt 0.005] [--ncell 64] [--reps 30] [--bank 0|1] [--melt 0]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import azplugins_amd as azp
from azplugins_amd import synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--kT", type=float, default=1.0)
ap.add_argument("--dt", type=float, default=0.005)
ap.add_argument("--ncell", type=int, default=64)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--bank", type=int, default=-1, help="-1: the library's own policy, 0 / 1: bank-aware rows off / on")
ap.add_argument("--melt", type=int, default=0, help="NVE steps before the recorded cycle (0 = start from the lattice)")
ap.add_argument("--fused", type=int, default=1, help="1: plan compiled straight from the cell list (default), 0: from the u32 list")
ap.add_argument("--only", type=int, default=-1, help="time only this step of the cycle (for PMC passes)")
args = ap.parse_args()

cfg = syn.config_north_star(args.ncell)
N = cfg["xyz"].shape[0]
sim = azp.Simulation(device="cuda:0", seed=1)
sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"]))
st = sim.state
nl = azp.nlist.Cell(buffer=cfg["r_buff"])
nl.fused = bool(args.fused)
pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=cfg["r_cut"])
pot.params[("A", "A")] = cfg["params"]
sim.operations.integrator = azp.Integrator(dt=args.dt, forces=[pot], methods=[azp.ConstantVolume()])
sim.operations.tuners.clear()
sim.run(0)
sim.thermalize_particle_momenta(args.kT, seed=7)
if args.melt:
    sim.run(args.melt)
    nl.compute(st, force=True)
    pot.compute(0)
builds0 = nl.num_builds
snaps = [st.pos.clone()]
while True:
    sim.run(1)
    if nl.num_builds != builds0:
        break
    snaps.append(st.pos.clone())
    if len(snaps) > 64:
        break


def timed(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pot.compute(0)
    e0.record()
    for _ in range(reps):
        pot.compute(0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# list + plan for snapshot 0
st.pos.copy_(snaps[0])
st.position_generation += 1
nl.compute(st, force=True)
if args.bank >= 0:
    pot._calls_since_plan = 1000 if args.bank else 0
pot.compute(0)
b1 = nl.num_builds
rows = []
for k, s in enumerate(snaps):
    if args.only >= 0 and k != args.only:
        continue
    st.pos.copy_(s)
    st.position_generation += 1
    pot.use_displacement_bound = True
    pot.compute(0)
    assert nl.num_builds == b1, "snapshot %d triggered a rebuild" % k
    bound = nl.displacement_bound(st)
    t_md = timed(args.reps)
    pot.use_displacement_bound = False
    t_whole = timed(args.reps)
    rows.append(dict(step=k, displacement_bound=bound, ms_md=t_md, ms_whole_rows=t_whole))
    print("step %2d  bound %.4f (%.2f x r_buff/2)  md %.4f ms   whole rows %.4f ms" % (k, bound, bound / (0.5 * cfg["r_buff"]), t_md, t_whole),
          flush=True)
mean_md = sum(r["ms_md"] for r in rows) / len(rows)
mean_whole = sum(r["ms_whole_rows"] for r in rows) / len(rows)
out = dict(N=N, kT=args.kT, dt=args.dt, melt=args.melt, steps_per_cycle=len(snaps), bank=args.bank, mean_ms_md=mean_md, mean_ms_whole_rows=mean_whole,
           mean_neighbors=nl.n_pairs / N, plan=pot.plan_info, rows=rows)
print(json.dumps(out))
