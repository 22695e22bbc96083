"""What one rank of an N-GPU weak-scaling run looks like, on ONE GPU: builds rank 0's
domain (local + ghost particles straight from the replicated synthetic configuration,
no communication), its neighbor list and tile plan, and times the interior and
boundary launches of bench.py's step. Answers: does the plan stay valid for the slab-
shaped boundary tiles, which LDS variant do the two launches get, what does the force
part of a step cost per rank.

    python tools/dd_plan_check.py [--worlds 2,4,8]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import azplugins_amd as azp
from azplugins_amd import synthetic as syn
from azplugins_amd.decomposition import Decomposition, HaloExchange, build_rank_state, choose_grid

ap = argparse.ArgumentParser()
ap.add_argument("--worlds", default="2,4,8")
ap.add_argument("--tpp", type=int, default=0, help="threads per particle of the tile kernel (0: the library's choice; 2, 4: list-based plan, smaller tiles)")
ap.add_argument("--strong", action="store_true", help="the north star's own 2^20 particles cut into `world` sub-boxes (default: 2^20 per rank)")
args = ap.parse_args()


def timed(fn, reps=50):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for world in [int(w) for w in args.worlds.split(",")]:
    grid = choose_grid(world, np.ones(3))
    cfg = syn.config_north_star(64) if args.strong else syn.config_north_star(tuple(64 * g for g in grid))
    decomp = Decomposition(cfg["L"], world, cfg["r_cut"] + cfg["r_buff"], grid=grid)
    dom, state = build_rank_state(cfg, decomp, 0, "cuda:0")
    halo = HaloExchange(dom, "cuda:0")
    sim = azp.Simulation(device="cuda:0", seed=1)
    sim.state = state
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=cfg["r_cut"])
    pot.params[("A", "A")] = cfg["params"]
    pot.threads_per_particle = args.tpp
    sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot])
    sim.run(0)
    n_int, n_bnd = dom.n_interior, dom.N_local - dom.n_interior
    t_all = timed(lambda: pot.compute(0))
    t_int = timed(lambda: pot.compute(0, particle_range=(0, n_int)))
    lds_int = azp._lib.last_launch()["lds_bytes"]
    t_bnd = timed(lambda: pot.compute(0, particle_range=(n_int, n_bnd)))
    lds_bnd = azp._lib.last_launch()["lds_bytes"]
    t_pack = timed(lambda: halo.pack(state.pos))
    # the two launches on two streams, split on a tile boundary so that they write disjoint rows
    n_al = (n_int // 256) * 256
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    main = torch.cuda.current_stream()

    def both():
        sa.wait_stream(main)
        sb.wait_stream(main)
        with torch.cuda.stream(sa):
            pot.compute(0, particle_range=(0, n_al))
        with torch.cuda.stream(sb):
            pot.compute(0, particle_range=(n_al, dom.N_local - n_al))
        main.wait_stream(sa)
        main.wait_stream(sb)

    t_conc = timed(both)
    info = pot.plan_info
    print("world %d grid %s: N_local %d (interior %d, boundary %d), ghosts %d; plan valid=%d max_stage=%d; one launch %.4f ms; "
          "interior %.4f ms (LDS %d B) + boundary %.4f ms (LDS %d B) = %.4f ms, on two streams %.4f ms; pack %.4f ms; "
          "halo %.2f MB sent per step"
          % (world, grid, dom.N_local, n_int, n_bnd, dom.n_ghost, info["valid"], info["max_stage"], t_all, t_int, lds_int,
             t_bnd, lds_bnd, t_int + t_bnd, t_conc, t_pack, halo.send_idx.numel() * 32 / 1e6), flush=True)
    del sim, pot, nl, state, dom, halo
    torch.cuda.empty_cache()
