#!/usr/bin/env python3
"""A/B timing of the PerturbedLJ tile kernel over the MD rebuild cycle bench.py times, in ONE process:
the variants alternate pass by pass (a pass = the 8 cycle states x --per-state launches, back to back), so
clock and thermal drift hit both alike.

    python3 tools/ab_cycle.py --passes 12 local=1 local=0
    variants: comma-separated key=value settings of the potential / environment-free knobs:
      local=0|1   per-particle displacements passed or not
      bound=0|1   displacement bound passed or not (whole rows)
      phases=0|1  row phases of the tile kernel (azp_tuning_set)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--passes", type=int, default=10)
    ap.add_argument("--per-state", type=int, default=10)
    ap.add_argument("--settle-ms", type=float, default=150.0)
    args = ap.parse_args()
    import torch

    import azplugins_amd as azp
    import bench

    cfg = bench.make_workload("ns")
    N = cfg["xyz"].shape[0]
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.create_state_from_snapshot(azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"]))
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=cfg["r_cut"], mode="none")
    pot.params[("A", "A")] = cfg["params"]
    sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot])
    sim.run(0)
    st = sim.state
    x0 = st.pos.clone()
    snaps = bench.record_md_cycle(sim, nl, 1.0, seed=7)
    st.pos = x0
    st.position_generation += 1
    pot.plan_bank_order = False
    nl.compute(st, force=True)
    pot.compute(0)
    bounds, disps = [], []
    for x in snaps:
        st.pos = x
        st.position_generation += 1
        pot.compute(0)
        bounds.append(nl.displacement_bound(st))
        d = nl.displacements(st)
        disps.append(d.clone() if d is not None else None)
    n_states = len(snaps)

    def apply(v):
        for kv in v.split(","):
            k, val = kv.split("=")
            if k == "local":
                pot.use_local_displacement = bool(int(val))
            elif k == "bound":
                pot.use_displacement_bound = bool(int(val))
            elif k == "phases":
                azp._lib.lib().azp_tuning_set(1, int(val))
            else:
                raise SystemExit("unknown knob %r" % k)

    def one_pass():
        for k in range(n_states):
            st.pos = snaps[k]
            st.position_generation += 1
            nl.assume_displacement(st, bounds[k], per_particle=disps[k])
            for _ in range(args.per_state):
                pot.compute(0)

    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < args.settle_ms:
        one_pass()
        torch.cuda.synchronize()
    res = {v: [] for v in args.variants}
    for p in range(args.passes):
        for v in args.variants:
            apply(v)
            one_pass()  # (untimed: the variant's own steady state)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            one_pass()
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / (n_states * args.per_state))
    for v in args.variants:
        a = np.array(res[v])
        print("%-24s mean %.4f ms  median %.4f  min %.4f  max %.4f  (%d passes)" % (v, a.mean(), np.median(a), a.min(), a.max(), len(a)))


if __name__ == "__main__":
    main()
