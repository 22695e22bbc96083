"""C4 (DPD) / C5 (TwoPatchMorse) at full size: tile-staged kernel vs the generic kernel, same inputs.

    python tools/xtiled_probe.py [c4|c5] [--reps 30]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import azplugins_amd as azp
from azplugins_amd import synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("which", nargs="?", default="c4")
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--settle-ms", type=float, default=80.0)
ap.add_argument("--sort", action="store_true", help="Hilbert-sort the particles first, as an MD run does every 200 steps (the synthetic "
                                                     "configurations come cell by cell: looser tiles)")
args = ap.parse_args()
cfg = syn.config_dpd() if args.which == "c4" else syn.config_tpm()
sim = azp.Simulation(device="cuda:0", seed=cfg.get("seed", 1))
sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], velocity=cfg.get("vel"), tag=cfg.get("tag"),
                                                        orientation=cfg.get("orientation")))
nl = azp.nlist.Cell(buffer=cfg["r_buff"])
if args.which == "c4":
    pot = azp.pair.DPDGeneralWeight(nlist=nl, kT=cfg["kT"], default_r_cut=cfg["r_cut"])
else:
    pot = azp.pair.TwoPatchMorse(nlist=nl, default_r_cut=cfg["r_cut"], mode="shift")
pot.params[("A", "A")] = cfg["params"]
sim.operations.integrator = azp.Integrator(dt=cfg.get("dt", 0.005), forces=[pot])
pot.use_plan = True
sim.operations.tuners.clear()
if args.sort:
    azp.ParticleSorter().sort(sim)
sim.run(0)


def settle(ms=args.settle_ms):
    # time at the sustained clock (profiles/r03_clock_transient.md), as bench.py does
    import time

    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(20):
            pot.compute(0)
        torch.cuda.synchronize()


def timed():
    settle()
    pot.compute(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        pot.compute(0)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.reps


t_plan = timed()
info = pot.plan_info
pot.use_displacement_bound = False
t_whole = timed()
pot.use_plan = False
t_gen = timed()
# the entry the reference's kernel driver forwards to (gpu_compute_dpd_forces / gpu_compute_pair_aniso_forces ->
# azp_dpd_forces_general_weight / azp_aniso_forces_two_patch_morse) with libazp's own plan cache: list check + tile kernel
import ctypes as C
import time

nl.fused = False          # HOOMD hands its own u32 list to the entry point: build one
pot.use_plan = True
pot._plan_builds = None
nl.compute(sim.state, force=True)
pot.compute(0)
a = pot._pair_args()
a.r_list_max = 0.0        # pair_args_t has no such field
a.has_displacement_bound = 0
a.flags = 0
a.threads_per_particle = 0
cargs = pot._wrap_args(a, 0)
lib = azp._lib.lib()
lib.azp_pair_auto_plan_clear()
fn = getattr(lib, pot._entry)
params = pot._tables["params"].data_ptr()
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    azp._lib.check(fn(C.byref(cargs), params, stream), pot._entry)
torch.cuda.synchronize()
t0 = time.perf_counter()
while (time.perf_counter() - t0) * 1e3 < args.settle_ms:
    for _ in range(20):
        azp._lib.check(fn(C.byref(cargs), params, stream), pot._entry)
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.reps):
    azp._lib.check(fn(C.byref(cargs), params, stream), pot._entry)
torch.cuda.synchronize()
t_entry = (time.perf_counter() - t0) / args.reps * 1e3
st = azp._lib.auto_plan_stats()
lib.azp_pair_auto_plan_clear()
print("%s: HOOMD-signature entry with the plan cache %.4f ms per call on the host (reuses %d, generic fallbacks %d)" % (
    args.which, t_entry, st["reuses"], st["generic_fallbacks"]))
print("%s: tile-staged %.4f ms (bound 0) / %.4f ms (whole rows), generic %.4f ms; plan %s; launch %s" % (
    args.which, t_plan, t_whole, t_gen, {k: info[k] for k in ("valid", "lds_slots", "max_stage", "tile_size")}, azp._lib.last_launch()))
