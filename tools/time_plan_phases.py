"""Plan-build timing against a library built with -DPLAN_ABLATE=1|2|3 (AZP_LIB_PATH): phase breakdown.
The product run uses the generic kernel (an ablated library leaves the plan incomplete)."""
import sys, os, time
sys.path.insert(0, "/root/repo")
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
pot.use_plan = False
sim.run(0)
a = pot._pair_args()
from azplugins_amd import _lib
plan = _lib.PairPlan()
stream = torch.cuda.current_stream().cuda_stream
ts=[]
for rep in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    try:
        plan.build(a, stream)
    except Exception as e:
        pass
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("AZP_LIB_PATH","cur"), " ".join("%.3f" % t for t in ts))
