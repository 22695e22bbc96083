"""Energy conservation of translational + rotational NVE with TwoPatchMorse, versus the time step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

R_CUT = float(os.environ.get('R_CUT', '4.0'))

import azplugins_amd as azp
from azplugins_amd import synthetic as syn

cfg = syn.config_tpm(10, 10, 10)
n = cfg["xyz"].shape[0]
tag = np.arange(n, dtype=np.uint64)
vel = np.stack([syn.normal(77, tag, c) for c in range(3)], axis=1) * np.sqrt(0.05)
vel -= vel.mean(axis=0)
for rot in (True, False):
    for dt in (0.004, 0.002, 0.001, 0.0005):
        snap = azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"], velocity=vel, orientation=cfg["orientation"],
                                        moment_inertia=np.tile([0.1, 0.1, 0.1], (n, 1)))
        sim = azp.Simulation(device="cuda:0", seed=1)
        sim.create_state_from_snapshot(snap)
        nl = azp.nlist.Cell(buffer=0.4)
        tpm = azp.pair.TwoPatchMorse(nlist=nl, default_r_cut=R_CUT, mode="none")
        tpm.params[("A", "A")] = dict(M_d=1.0, M_r=0.25, r_eq=1.1, omega=5.0, alpha=0.4, repulsion=True)
        sim.operations.integrator = azp.Integrator(dt=dt, forces=[tpm], methods=[azp.ConstantVolume()], integrate_rotational_dof=rot)
        sim.operations.tuners.clear()
        sim.run(0)

        def total():
            v = sim.state.vel[:n]
            k_t = 0.5 * float((v[:, 3] * (v[:, :3] ** 2).sum(dim=1)).sum().item())
            return tpm.energy, k_t, sim.rotational_kinetic_energy()

        u0, kt0, kr0 = total()
        sim.run(int(round(1.0 / dt)))
        u1, kt1, kr1 = total()
        print("rot=%s dt=%.4f: U %.3f -> %.3f  Kt %.3f -> %.3f  Kr %.3f -> %.3f  dE = %.5f" % (
            rot, dt, u0, u1, kt0, kt1, kr0, kr1, (u1 + kt1 + kr1) - (u0 + kt0 + kr0)), flush=True)
