"""Times azp_pair_plan_build (first build and steady-state rebuilds) and the
nlist build on the north-star workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import azplugins_amd as azp
from azplugins_amd import synthetic as syn

cfg = syn.config_north_star(64)
sim = azp.Simulation(device="cuda:0", seed=1)
sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"]))
nl = azp.nlist.Cell(buffer=0.4)
nl.fused = False  # these tools time the list-based plan compiler
pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=3.0)
pot.params[("A", "A")] = cfg["params"]
sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot])
sim.run(0)
a = pot._pair_args()
stream = torch.cuda.current_stream().cuda_stream
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pot._plan.build(a, stream)
    torch.cuda.synchronize(); print("plan build %d: %.3f ms" % (rep, (time.perf_counter() - t0) * 1e3), pot.plan_info)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    nl.compute(sim.state, force=True)
    torch.cuda.synchronize(); print("nlist build %d: %.3f ms" % (rep, (time.perf_counter() - t0) * 1e3))
