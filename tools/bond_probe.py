"""C3's DoubleWell bond kernel alone (N = 1,048,576, 1,015,808 bonds): launch time, for rocprofv3 passes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import azplugins_amd as azp
from azplugins_amd import synthetic as syn

cfg = syn.config_chains()
sim = azp.Simulation(device="cuda:0", seed=1)
sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], bonds=cfg["bonds"]))
dw = azp.bond.DoubleWell()
dw.params["A-A"] = cfg["bond_params"]
sim.operations.integrator = azp.Integrator(dt=0.001, forces=[dw])
sim.run(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
dw.compute(0)
e0.record()
for _ in range(reps):
    dw.compute(0)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
N = cfg["xyz"].shape[0]
nb = 2.0 * len(cfg["bonds"]) / N
b_alg = 32 + 4 + 12 * nb + 32
print("bond kernel: %.4f ms per launch, %.0f GB/s algorithmic (%.1f B per particle)" % (ms, b_alg * N / ms / 1e6, b_alg))
