"""Host-side cost of one Pair.compute() call (ctypes struct fill + launch), measured on a
system small enough that the GPU is never the limit."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import azplugins_amd as azp
from azplugins_amd import synthetic as syn

cfg = syn.config_north_star(8)
sim = azp.Simulation(device="cuda:0", seed=1)
sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"]))
nl = azp.nlist.Cell(buffer=0.4)
pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=3.0)
pot.params[("A", "A")] = cfg["params"]
sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot])
sim.run(0)
for label, fn in (("compute()", lambda: pot.compute(0)), ("compute(range)", lambda: pot.compute(0, particle_range=(0, 1024)))):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        fn()
    torch.cuda.synchronize()
    print("%-16s %.1f us per call (N=%d)" % (label, 1e6 * (time.perf_counter() - t0) / 2000, cfg["xyz"].shape[0]))
