#!/usr/bin/env python3
"""Benchmark of the force-compute hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ns|c2|...]

Metric (BASELINE.json): particle-steps/s of the PerturbedLennardJones pair
force at N=1,048,576, rho*=0.8, r_cut=3.0 (FP64, full neighbor list, buffer
0.4), plus the achieved algorithmic HBM GB/s against the MI355X roofline.

A "step" is one pass of the force kernel(s) over the whole system on a static
neighbor list, inputs resident in HBM. Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # spec peak, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
HBM_COPY_GBS = 6290.0      # measured float4 copy on the same guide


def alg_bytes_per_particle(mean_neigh, S=8, extra=0):
    """4S pos + 4 n_neigh + 8 head + 4<n> nlist + 4S force (+ extra)."""
    return 4 * S + 4 + 8 + 4.0 * mean_neigh + 4 * S + extra


def make_workload(name):
    from azplugins_amd import synthetic as syn

    if name == "ns":
        return syn.config_north_star(64)
    if name == "ns-small":
        return syn.config_north_star(16)
    if name == "c2":
        return syn.config_plj_sc(64)
    if name.startswith("ns-x"):  # ns-x2 / ns-x4 / ns-x8: the north star repeated along z (scaling studies)
        return syn.config_north_star((64, 64, 64 * int(name[4:])))
    raise SystemExit("unknown workload %r" % name)


def measured_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/r*_traffic.json), or None. bench.py cannot collect PMC counters
    itself; the JSON names the passes it came from."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            d = json.load(open(f)).get(kernel)
        except (OSError, ValueError):
            continue
        if d and d.get("workload", "").lower() == workload.lower():
            return (d["FETCH_SIZE_KiB"] * d.get("fetch_correction", 1.0) + d["WRITE_SIZE_KiB"]) * 1024.0, os.path.basename(f)
    return None, None


def displace_particles(state, amplitude, seed):
    """Move every local particle by a hashed random vector (uniform direction, length
    uniform in [0, amplitude]) and wrap it back into the box: the state of an MD run
    part-way between two neighbor-list rebuilds."""
    import torch

    from azplugins_amd import synthetic as syn

    n = state.N
    tag = np.arange(n, dtype=np.uint64)
    v = np.stack([syn.normal(seed, tag, c) for c in range(3)], axis=1)
    v *= (amplitude * syn.u01(seed, tag, 7) / np.linalg.norm(v, axis=1))[:, None]
    x = state.pos[:n, :3].cpu().numpy() + v
    x = syn.wrap(x, state.box.L)
    state.pos[:n, :3] = torch.from_numpy(x).to(state.pos.device)
    state.position_generation += 1


def cpu_baseline(workload, reps=5):
    """HOOMD-equivalent CPU loop restated (oracle): half neighbor list, third-law
    scatter, FP64, ONE core (HOOMD's per-rank CPU execution model), timed on
    this host. Bounded sample: the same lattice, density, potential and cutoff
    at 1/8 of the particle count (per-particle cost is size-independent)."""
    import oracle
    from azplugins_amd import synthetic as syn

    cfg = syn.config_north_star(32) if workload.startswith("ns") else syn.config_plj_sc(40)
    pos = syn.pos4(cfg["xyz"])
    box = oracle.make_box(cfg["L"])
    params = oracle.pack_pair_params(cfg["potential"], cfg["params"])
    nl = oracle.build_nlist(pos, box, cfg["r_cut"] + cfg["r_buff"], half=True)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        oracle.pair_forces(cfg["potential"], pos, box, nl, params, cfg["r_cut"], mode="none", half=True)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    N = pos.shape[0]
    out = dict(value=N / t, unit="particle-steps/s", cores=1, kind="port",
               sample="%s: same lattice/density/potential, N=%d (1/8 of the workload), half list, median of %d "
                      "reps; oracle = HOOMD-equivalent loop restated, not the HOOMD binary" % (cfg["name"], N, reps))
    # best-effort all-core figure (OpenMP over particles, full list), on this process's CPU share
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    ncores = min(ncores, 16)
    nl_full = oracle.build_nlist(pos, box, cfg["r_cut"] + cfg["r_buff"], half=False)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        oracle.pair_forces(cfg["potential"], pos, box, nl_full, params, cfg["r_cut"], mode="none", nthreads=ncores)
        ts.append(time.perf_counter() - t0)
    out["all_cores"] = dict(value=N / float(np.median(ts)), cores=ncores)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ns")
    ap.add_argument("--tpp", type=int, default=0, help="threads per particle (0 = library heuristic)")
    ap.add_argument("--block-size", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="none", choices=["none", "shift", "xplor"])
    ap.add_argument("--no-plan", action="store_true", help="generic kernel only (no LDS-staged tile plan)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1 GPUs: weak = 2^20 particles per GPU (default), strong = 2^20 in total")
    ap.add_argument("--displace", type=float, default=0.0,
                    help="sensitivity run: after the neighbor list and plan are built, move every particle by a hashed "
                         "random vector of length <= DISPLACE * r_buff / 2 (0 = the snapshot the list was built for, as "
                         "the metric is defined; 1 = the moment before the next rebuild)")
    ap.add_argument("--sort-rows", action="store_true", help="experiment: sort every neighbor row by index before planning")
    ap.add_argument("--cycle-report", action="store_true",
                    help="after the timed region, also time the kernel without displacement information and with every "
                         "particle displaced by 0.5 / 0.9 x r_buff/2 (same list and plan); reported in config")
    ap.add_argument("--no-displacement-bound", action="store_true",
                    help="do not tell the planned kernel how far particles moved since the list was built (it then walks "
                         "whole rows, Verlet-buffer entries included)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("AZP_BENCH_ONE_DEVICE") == "1":  # rehearsal: all ranks share GPU 0 (with AZP_DIST_BACKEND=gloo)
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the force path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank

    if world > 1 or os.environ.get("AZP_BENCH_FORCE_DD") == "1":  # the env var rehearses the N>1 code path on one GPU
        from azplugins_amd import decomposition

        return decomposition.bench_main(args, rank, world, local_rank)

    import azplugins_amd as azp

    cfg = make_workload(args.workload)
    N = cfg["xyz"].shape[0]
    snap = azp.Snapshot.from_arrays(cfg["xyz"], cfg["L"])
    sim = azp.Simulation(device=dev, seed=1)
    sim.create_state_from_snapshot(snap)
    nl = azp.nlist.Cell(buffer=cfg["r_buff"])
    pot = azp.pair.PerturbedLennardJones(nlist=nl, default_r_cut=cfg["r_cut"], mode=args.mode)
    pot.params[("A", "A")] = cfg["params"]
    pot.threads_per_particle = args.tpp
    pot.block_size = args.block_size
    pot.use_plan = not args.no_plan
    pot.use_displacement_bound = not args.no_displacement_bound
    sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot])
    sim.run(0)  # attaches, builds the neighbor list (and the tile plan) on the GPU, first force evaluation
    if args.sort_rows:
        row = torch.repeat_interleave(torch.arange(N, device=dev, dtype=torch.int64), nl.n_neigh.to(torch.int64))
        key = (row << 32) | nl.nlist[: nl.size].to(torch.int64)
        nl.nlist[: nl.size] = (torch.sort(key).values & 0xFFFFFFFF).to(torch.int32)
        nl.num_builds += 1  # forces a plan rebuild
        del row, key
        pot.compute(0)
    mean_neigh = nl.n_pairs / N
    if args.displace > 0.0:
        displace_particles(sim.state, args.displace * 0.5 * cfg["r_buff"], seed=11)
        pot.compute(0)
        assert nl.num_builds == 1, "the displacement must not trigger a neighbor-list rebuild"

    for _ in range(args.warmup):
        pot.compute(0)
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        pot.compute(0)
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
    ms_per_step = wall * 1e3 / args.steps

    # the same list part-way through a rebuild cycle (outside the timed region): every particle
    # displaced by <= 0.5 x r_buff/2, then by <= 0.9 x r_buff/2; and with no displacement information
    def timed(reps=40):
        pot.compute(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            pot.compute(0)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    # always: a handful of launches over whole rows (no displacement information), so that both
    # figures are in every JSON line; few enough not to move the profiler's per-kernel average
    whole_rows_ms = None
    if pot.use_displacement_bound and args.displace == 0.0 and not args.cycle_report:
        pot.use_displacement_bound = False
        whole_rows_ms = timed(reps=5)
        pot.use_displacement_bound = True

    cycle = {}
    if args.cycle_report and args.displace == 0.0 and not args.sort_rows:
        pot.use_displacement_bound = False
        cycle["no_displacement_information_ms"] = timed()
        pot.use_displacement_bound = not args.no_displacement_bound
        x0 = sim.state.pos.clone()
        for f in (0.5, 0.9):
            sim.state.pos.copy_(x0)
            displace_particles(sim.state, f * 0.5 * cfg["r_buff"], seed=11)
            cycle["displaced_%.1f_x_half_buffer_ms" % f] = timed()
        assert nl.num_builds == 1
        sim.state.pos.copy_(x0)
        sim.state.position_generation += 1

    value = N * args.steps / wall
    b_alg = alg_bytes_per_particle(mean_neigh)
    achieved = b_alg * N / (kernel_ms * 1e-3) / 1e9
    launch = azp._lib.last_launch()
    kernel_name = ("azp::pair_forces_tiled_kernel<EvalPLJ>" if (pot.plan_info or {}).get("valid")
                   else "azp::pair_forces_kernel<EvalPLJ>")
    traffic, traffic_src = measured_traffic(kernel_name, cfg["name"])
    out = {
        "metric": "particle-steps/sec, PerturbedLennardJones pair force",
        "value": value,
        "unit": "particle-steps/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "%s: PerturbedLennardJones N=%d rho*=0.8 r_cut=%.1f buffer=%.1f mode=%s, jittered %s lattice, "
                        "full neighbor list <n>=%.2f" % (cfg["name"], N, cfg["r_cut"], cfg["r_buff"], args.mode,
                                                        "FCC" if args.workload.startswith("ns") else "SC", mean_neigh),
            "N": N,
            "mean_neighbors": mean_neigh,
            "displacement_since_list_build": "every particle moved by <= %.3g (= %.2f x r_buff/2)"
                                             % (args.displace * 0.5 * cfg["r_buff"], args.displace),
            "displacement_bound_passed": bool(pot.use_displacement_bound),
            "note": "static list (the metric's definition): the kernel is told that no particle moved since the list was "
                    "built and stops each row before its Verlet-buffer entries (exact). --cycle-report times the same "
                    "list mid-cycle; profiles/r01d_cycle.json holds that run.",
            "kernel_ms_whole_rows_no_displacement_information": whole_rows_ms,
            "kernel_ms_elsewhere_in_a_rebuild_cycle": cycle,
            "launch": launch,
            "tile_plan": pot.plan_info,
            "parallelism": "1 GPU",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_of_measured_copy_peak": achieved / HBM_COPY_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "kernel": kernel_name,
            "kernel_ms": kernel_ms,
            "algorithmic_bytes_per_particle": b_alg,
        },
    }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload)
        out["cpu_baseline"]["gpu_over_cpu_1core"] = value / out["cpu_baseline"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
