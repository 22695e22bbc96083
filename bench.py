#!/usr/bin/env python3
"""Benchmark of the force-compute hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload ns|c2|...]

Metric (BASELINE.json): particle-steps/s of the PerturbedLennardJones pair
force at N=1,048,576, rho*=0.8, r_cut=3.0 (FP64, full neighbor list, buffer
0.4), plus the achieved algorithmic HBM GB/s against the MI355X roofline.

A "step" is one pass of the force kernel(s) over the whole system on a static
neighbor list (built once, never rebuilt in the timed region), inputs resident in
HBM. The positions the kernel sees cycle through the steps of one rebuild cycle of
a real MD run (the list is built for step 0; HOOMD's distance check would rebuild
it after the last one), so the figure is the mean over a cycle and not the
best-case step right after a rebuild; --static times that step alone. Prints ONE
JSON line on rank 0.
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


def cpu_baseline(workload, reps=3, gpu_forces_by_tag=None):
    """HOOMD-equivalent CPU loop restated (oracle): half neighbor list, third-law
    scatter, FP64, ONE core (HOOMD's per-rank CPU execution model), timed on
    this host on the workload itself (the full N; ~1 s per repetition)."""
    import oracle

    cfg = make_workload(workload)
    pos = syn_pos4(cfg)
    box = oracle.make_box(cfg["L"])
    params = oracle.pack_pair_params(cfg["potential"], cfg["params"])
    nl = oracle.build_nlist(pos, box, cfg["r_cut"] + cfg["r_buff"], half=True)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f_cpu = oracle.pair_forces(cfg["potential"], pos, box, nl, params, cfg["r_cut"], mode="none", half=True)
        ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    N = pos.shape[0]
    checked = None
    if gpu_forces_by_tag is not None:
        # the oracle is the checker here, never the thing measured: the forces the GPU path produced for
        # cycle step 0 (= the workload's snapshot) against the CPU loop's, every particle
        checked = float(np.abs(gpu_forces_by_tag - f_cpu).max() / np.abs(f_cpu).max())
        if not (checked < 1e-10):
            raise SystemExit("bench.py: GPU forces differ from the CPU oracle on the workload snapshot: %g" % checked)
    out = dict(value=N / t, unit="particle-steps/s", cores=1, kind="port",
               sample="%s: the workload itself, N=%d, half list (built outside the timed region), median of %d "
                      "repetitions of the force loop; oracle = HOOMD-equivalent loop restated, not the HOOMD binary" % (cfg["name"], N, reps))
    if checked is not None:
        out["gpu_vs_cpu_max_abs_err_over_max_force"] = checked
    del nl
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


def syn_pos4(cfg):
    from azplugins_amd import synthetic as syn

    return syn.pos4(cfg["xyz"])


def committed_counter(kernel, workload, name):
    """Mean per-launch value of a PMC counter from the committed rocprofv3 passes
    (profiles/r*_counters.json), or None."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")), reverse=True):
        try:
            d = json.load(open(f)).get(kernel)
        except (OSError, ValueError):
            continue
        if d and d.get("workload", "").lower() == workload.lower() and name in d:
            return float(d[name]), os.path.basename(f)
    return None, None


def record_md_cycle(sim, nl, kT, seed):
    """Positions of one neighbor-list rebuild cycle of a real NVE run: Maxwell velocities at
    kT on the workload's snapshot, velocity Verlet until HOOMD's distance check (a particle
    moved farther than r_buff / 2) asks for the next rebuild. Returns the position tensors
    after steps 0, 1, ... (step 0 = the snapshot the list is built for)."""
    import azplugins_amd as azp

    st = sim.state
    integ = sim.operations.integrator
    integ.methods = [azp.ConstantVolume()]
    sim.operations.tuners.clear()
    sim.thermalize_particle_momenta(kT, seed=seed)
    builds = nl.num_builds
    snaps = [st.pos.clone()]
    while len(snaps) < 64:
        sim.run(1)
        if nl.num_builds != builds:
            break
        snaps.append(st.pos.clone())
    integ.methods = []
    return snaps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="ns")
    ap.add_argument("--tpp", type=int, default=0, help="threads per particle (0 = library heuristic)")
    ap.add_argument("--block-size", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-figures", action="store_true",
                    help="only the timed region (profiler passes: every launch of the kernel belongs to the cycle)")
    ap.add_argument("--mode", default="none", choices=["none", "shift", "xplor"])
    ap.add_argument("--no-plan", action="store_true", help="generic kernel only (no LDS-staged tile plan)")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1 GPUs: strong = 2^20 particles in total (default, the north star's criterion), weak = 2^20 per GPU")
    ap.add_argument("--static", action="store_true",
                    help="time only the snapshot the list was built for (displacement bound 0: round 1's headline state) "
                         "instead of a rebuild cycle of an MD run")
    ap.add_argument("--settle-ms", type=float, default=80.0,
                    help="untimed run-in before the W warmup steps: the same launches on the same cycle states, back to back, for this "
                         "many milliseconds, so that the timed region sits at the clock the chip SUSTAINS under this kernel (as "
                         "in an MD run of millions of steps) and not in the power controller's transient: from idle the first "
                         "~1 ms runs at boost clock, launches 15-50 run up to 35 %% slower, and the clock settles over ~50 ms "
                         "(profiles/r03_clock_transient.md). 0 = none")
    ap.add_argument("--no-verify", action="store_true", help="skip the check of the timed states' forces after the timed region")
    ap.add_argument("--kT", type=float, default=1.0, help="temperature of the Maxwell velocities that drive the recorded MD cycle")
    ap.add_argument("--bank-order", type=int, default=0,
                    help="1: bank-aware row order in the tile plan (what a list that lives >= 50 force calls gets); "
                         "default 0, as in an MD run that rebuilds every ~8 steps")
    ap.add_argument("--displace", type=float, default=0.0,
                    help="sensitivity run: after the neighbor list and plan are built, move every particle by a hashed "
                         "random vector of length <= DISPLACE * r_buff / 2 (0 = the snapshot the list was built for, as "
                         "the metric is defined; 1 = the moment before the next rebuild)")
    ap.add_argument("--no-fused-plan", action="store_true",
                    help="build HOOMD's u32 neighbor list and compile the tile plan from it (default: plan straight from the cell list)")
    ap.add_argument("--balance", type=int, default=-1,
                    help="1 / 0: rows of a tile to its lanes by in-range length (balanced plans) on / off; default: the class's own choice")
    ap.add_argument("--sort-rows", action="store_true", help="experiment: sort every neighbor row by index before planning")
    ap.add_argument("--local-bound", action="store_true",
                    help="also give the kernel the per-particle displacements of the distance check (a tile then stops its rows by the "
                         "displacements of its own particles; off by default: +1 %% on this workload, tools/ab_cycle.py)")
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
    if args.balance >= 0:
        pot._plan_balance = bool(args.balance)
    pot.use_displacement_bound = not args.no_displacement_bound
    sim.operations.integrator = azp.Integrator(dt=0.005, forces=[pot])
    if args.sort_rows or args.no_fused_plan:
        nl.fused = False  # HOOMD-format list first, plan compiled from it
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

    # ---- the states the kernel is timed on: one rebuild cycle of a real MD run ----
    # (args.static: the round-1 figure, the snapshot the list was built for, no particle moved)
    st = sim.state
    r_buff = cfg["r_buff"]
    pot.use_local_displacement = bool(args.local_bound)
    if args.static or args.displace > 0.0 or args.sort_rows:
        snaps, bounds = [st.pos], [nl.displacement_bound(st)]
        disps = [nl.displacements(st)]
    else:
        x0 = st.pos.clone()
        snaps = record_md_cycle(sim, nl, args.kT, seed=7)
        # list + plan for step 0, built as an MD run builds them (rows of fixed capacity; the
        # bank-aware row order only pays for lists that live >= 50 force calls: off)
        st.pos = x0
        st.position_generation += 1
        pot.plan_bank_order = bool(args.bank_order)
        nl.compute(st, force=True)
        pot.compute(0)
        builds = nl.num_builds
        bounds, disps = [], []
        for x in snaps:  # the distance check of every step, as the MD run made it
            st.pos = x
            st.position_generation += 1
            pot.compute(0)
            assert nl.num_builds == builds, "a recorded step triggered a rebuild"
            bounds.append(nl.displacement_bound(st))
            d = nl.displacements(st)
            disps.append(d.clone() if d is not None else None)  # (step 0 runs no check: no array, global bound 0)
    n_states = len(snaps)

    def set_state(k):
        st.pos = snaps[k]
        st.position_generation += 1
        # known from the recorded run (the maximum and every particle's own): no distance-check kernel in the timed loop
        nl.assume_displacement(st, bounds[k], per_particle=disps[k])

    # Timed order: the K steps walk through the cycle ONCE, steps // n_states consecutive launches on
    # each state (state of step k = k * n_states // K), so that every cycle step carries the same
    # weight and each launch reads positions that are as warm in the caches as they are in an MD
    # run, where the integrator has just written them. (Switching to another 32 MB snapshot at
    # every launch makes each launch stage positions that left the Infinity Cache eight launches
    # earlier: +7-9 % per launch; that order is timed after the region as a side figure.)
    def state_of(k, total):
        return min(k * n_states // max(total, 1), n_states - 1)

    def step(k, total, blocked=True):
        s = state_of(k, total) if blocked else k % n_states
        if step.current != s:
            set_state(s)
            step.current = s
        pot.compute(0)

    # run-in (untimed, before the W warmup steps): the cycle walked back to back until the chip's
    # power controller has settled (--settle-ms)
    def settle():
        n = 0
        if args.settle_ms > 0.0:
            per_pass = 8 * n_states
            t_settle = time.perf_counter()
            while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
                step.current = -1
                for k in range(per_pass):
                    step(k, per_pass)
                torch.cuda.synchronize()
                n += per_pass
        return n

    settle_launches = settle()
    step.current = -1
    for k in range(args.warmup):
        step(k, args.warmup)
    step.current = -1
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        step(k, args.steps)
    ev1.record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream
    ms_per_step = wall * 1e3 / args.steps

    # ---- check of what the timed region computed (outside it): the forces the timed launches left
    # behind for the LAST cycle state, and those of every other cycle state recomputed the same way
    # (same plan, same assumed displacement bound), must be bit-identical to a launch that is told
    # nothing about displacements and walks whole rows (the kernel's cutoff test is exact either way;
    # tests/test_gpu_full_size.py checks whole rows against the oracle at this size)
    verified = None
    f_state0 = None
    if not args.no_verify and pot.use_plan:
        timed_last = pot.force_tensor.clone()
        verified = True
        for k in reversed(range(n_states)):
            set_state(k)
            pot.compute(0)
            with_bound = pot.force_tensor.clone()
            if k == n_states - 1:
                verified = verified and bool((with_bound == timed_last).all())
            pot.use_displacement_bound = False
            pot.compute(0)
            pot.use_displacement_bound = not args.no_displacement_bound
            verified = verified and bool((pot.force_tensor == with_bound).all()) and bool(torch.isfinite(with_bound).all())
            if k == 0 and n_states > 1 and args.mode == "none":  # kept for the CPU oracle's check in cpu_baseline (by tag)
                f_state0 = np.empty((N, 4))
                f_state0[st.tag[:N].cpu().numpy().view(np.uint32)] = with_bound.cpu().numpy()
            del with_bound
        if not verified:
            raise SystemExit("bench.py: the forces of the timed states differ from the whole-row evaluation")
        step.current = -1

    # ---- side figures, outside the timed region (>= 40 launches each) ----
    def timed(reps=40):
        pot.compute(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            pot.compute(0)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    by_step = []
    if not args.no_side_figures:
        settle()  # the side figures too are taken at the sustained clock
    for k in range(n_states):
        if args.no_side_figures:
            break
        set_state(k)
        by_step.append(dict(step=k, displacement_bound=bounds[k], bound_over_half_buffer=bounds[k] / (0.5 * r_buff), kernel_ms=timed()))
    side = {}
    plan_info = pot.plan_info  # of the plan the timed region ran on (the side figures below recompile it)
    if plan_info and plan_info.get("valid"):
        plan_info["mean_chunks_core_sure_row"] = pot._plan.phase_chunks()
    if n_states > 1 and not args.no_side_figures:
        # the same K launches, switching to the next cycle step at every launch (cold positions)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        step.current = -1
        e0.record()
        for k in range(args.steps):
            step(k, args.steps, blocked=False)
        e1.record()
        torch.cuda.synchronize()
        side["cycle_mean_ms_switching_state_every_launch"] = e0.elapsed_time(e1) / args.steps
    set_state(0)
    if pot.use_displacement_bound and not args.sort_rows and args.displace == 0.0 and not args.no_side_figures:
        side["static_list_bound_0_ms"] = timed()  # round 1's headline state
        pot.use_displacement_bound = False
        side["whole_rows_no_displacement_information_ms"] = timed()
        pot.use_displacement_bound = True
        if n_states > 1:
            # round 1's synthetic mid-cycle states: EVERY particle displaced by a random vector of
            # length <= f r_buff / 2 (harsher than an MD step, where only the fastest particles get there)
            keep = st.pos
            for f in (0.5, 0.9):
                st.pos = snaps[0].clone()
                displace_particles(st, f * 0.5 * r_buff, seed=11)
                pot.compute(0)  # runs the distance check
                assert nl.num_builds == builds
                side["every_particle_displaced_%.1f_x_half_buffer_ms" % f] = timed()
            st.pos = keep
            st.position_generation += 1
            # the same cycle with the bank-aware row order (what a list that lives >= 50 calls gets;
            # only the list-based plan compiler orders rows that way)
            set_state(0)
            nl.fused = False
            nl.compute(st, force=True)
            pot.plan_bank_order = not bool(args.bank_order)
            pot._plan_builds = None
            alt = []
            for k in range(n_states):
                set_state(k)
                alt.append(timed(20))
            side["cycle_mean_ms_with_bank_order_%s" % ("on" if pot.plan_bank_order else "off")] = float(np.mean(alt))

    value = N * args.steps / wall
    b_alg = alg_bytes_per_particle(mean_neigh)
    achieved = b_alg * N / (kernel_ms * 1e-3) / 1e9
    launch = azp._lib.last_launch()
    kernel_name = ("azp::pair_forces_tiled_kernel<EvalPLJ>" if (plan_info or {}).get("valid")
                   else "azp::pair_forces_kernel<EvalPLJ>")
    traffic, traffic_src = measured_traffic(kernel_name, cfg["name"])
    valu, valu_src = committed_counter(kernel_name, cfg["name"], "SQ_INSTS_VALU")
    cycle_desc = ("static list: the snapshot the list was built for, displacement bound 0" if n_states == 1 else
                  "one neighbor-list rebuild cycle of an NVE run (Maxwell velocities kT=%.2f on the workload's snapshot, dt=0.005; "
                  "%d steps until HOOMD's distance check asks for a rebuild); the K timed steps walk through the cycle once, "
                  "K / %d consecutive launches on the positions of each cycle step with the displacement bound that step's "
                  "distance check returned; list and plan built once for step 0, not rebuilt"
                  % (args.kT, n_states, n_states))
    out = {
        "metric": "particle-steps/sec, PerturbedLennardJones pair force",
        "value": value,
        "unit": "particle-steps/s",
        "n_gpus": 1,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "verified": verified,
        "config": {
            "workload": "%s: PerturbedLennardJones N=%d rho*=0.8 r_cut=%.1f buffer=%.1f mode=%s, jittered %s lattice, "
                        "full neighbor list <n>=%.2f" % (cfg["name"], N, cfg["r_cut"], r_buff, args.mode,
                                                        "FCC" if args.workload.startswith("ns") else "SC", mean_neigh),
            "N": N,
            "mean_neighbors": mean_neigh,
            "states_timed": cycle_desc,
            "settle": "untimed run-in before the warmup: %d launches walking the same cycle back to back (%.0f ms), so that the "
                      "timed region runs at the sustained clock, not in the power controller's start-up transient" % (settle_launches, args.settle_ms),
            "kernel_ms_by_cycle_step": by_step,
            "kernel_ms_other_states": side,
            "plan_bank_order": bool(pot.plan_bank_order) if n_states > 1 else None,
            "displacement_bound_passed": bool(pot.use_displacement_bound),
            "per_particle_displacements_passed": bool(pot.use_local_displacement),
            "launch": launch,
            "tile_plan": plan_info,
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
            "note": "achieved = SURVEY 8(d) algorithmic bytes / mean launch time over the cycle. The bytes this kernel really moves "
                    "(traffic) are about half of that (16-bit compiled list); it is limited by FP64 VALU issue and LDS gather "
                    "cycles, not by HBM: see roofline_fp64_issue",
        },
    }
    if valu is not None:
        lane_ops = valu * 64.0 / (kernel_ms * 1e-3) / 1e12
        peak = 256 * 4 * 16 * 2.4e9 / 1e12  # 256 CUs x 4 SIMDs x 16 FP64 lanes per clock x 2.4 GHz (= 78.6 TFLOP/s FMA peak / 2)
        out["roofline_fp64_issue"] = {
            "bound": "fp64-issue", "achieved": lane_ops, "peak": peak, "unit": "T lane-instructions/s", "frac": lane_ops / peak,
            "valu_wave_instructions_per_launch": valu, "source": valu_src,
            "note": "VALU wave-instructions per launch (rocprofv3 --pmc SQ_INSTS_VALU, committed pass) x 64 lanes / the launch time "
                    "measured here, against the FP64 vector issue peak at the 2.4 GHz nominal clock; the chip holds ~1.5-1.8 GHz "
                    "under this load",
        }
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.workload, gpu_forces_by_tag=f_state0)
        out["cpu_baseline"]["gpu_over_cpu_1core"] = value / out["cpu_baseline"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
