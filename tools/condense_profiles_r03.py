#!/usr/bin/env python3
"""Condense the output of tools/profile_passes_r03.sh under gpurun_out/r03final into the files kept
under profiles/ (r03_*).

    python tools/condense_profiles_r03.py gpurun_out/r03final
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.abspath(sys.argv[1])
P = os.path.join(ROOT, "profiles")


def newest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1] if fs else None


def stats(d, match=("azp::",), top=14):
    f = newest(os.path.join(O, d, "*", "*_kernel_stats.csv"))
    if not f:
        return "(pass not run)"
    out = ["name,calls,total_ms,avg_us,pct,min_us,max_us"]
    for r in csv.DictReader(open(f)):
        if any(m in r["Name"] for m in match):
            out.append('"%s",%s,%.3f,%.1f,%s,%.1f,%.1f' % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                         r["Percentage"][:6], float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    return "\n".join(out[: top + 1])


def counters(d, kernel_substr):
    """mean per launch of every counter of a --pmc pass, for kernels whose name contains kernel_substr"""
    f = newest(os.path.join(O, d, "*", "*counter_collection.csv"))
    if not f:
        return {}, 0
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if kernel_substr in r["Kernel_Name"]:
            agg[(r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    per = collections.defaultdict(list)
    for (c, _), v in agg.items():
        per[c].append(v)
    n = max((len(v) for v in per.values()), default=0)
    return {c: sum(v) / len(v) for c, v in per.items()}, n


DRV = "--steps 20 --warmup 5"
md = ["# rocprofv3 --kernel-trace --stats summaries, round 3 (one MI355X), final binary of the round", ""]
md += ["## bench_stats: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-side-figures %s` (the driver's arguments; the launches are "
       "the recorded cycle, the 80 ms run-in, warm-up, the 20 timed steps and the verification)" % DRV, "", stats("bench_stats"), ""]
md += ["## md_stats: `rocprofv3 --kernel-trace --stats -- python3 tools/md_bench.py --steps 300` (north-star liquid, NVE)", "",
       stats("md_stats", match=("azp::", "rocprim", "copyBuffer", "at::native"), top=16), ""]
md += ["## c3_stats: `... -- python3 tools/md_bench.py --workload c3 --steps 300 --dt 0.002` (32,768 chains of 32: PerturbedLJ + DoubleWell, NVE)", "",
       stats("c3_stats", match=("azp::", "rocprim", "copyBuffer", "at::native"), top=16), ""]
md += ["## c4_stats: `... -- python3 tools/xtiled_probe.py c4` (DPD thermostat, N = 2,097,152)", "", stats("c4_stats"), ""]
md += ["## c5_stats: `... -- python3 tools/xtiled_probe.py c5` (TwoPatchMorse, N = 524,288)", "", stats("c5_stats"), ""]
md += ["## bond_stats: `... -- python3 tools/bond_probe.py` (C3's DoubleWell bonds alone)", "", stats("bond_stats"), ""]
md += ["## eval_stats: `... -- python3 tools/evaluator_probe.py` (every isotropic evaluator on the north-star geometry; tile kernel bound 0 / whole rows, generic)", "",
       stats("eval_stats", top=24), ""]
md += ["## entry_stats: `... -- python3 -m pytest tests/test_gpu_auto_plan.py -q -s -k plan_speed` (the HOOMD-signature entry: check + fold + tile kernel per call)", "",
       stats("entry_stats"), ""]
open(os.path.join(P, "r03_kernel_stats.md"), "w").write("\n".join(md))

for src, dst in (("bench_driver.json", "r03_bench.json"), ("bench_default.json", "r03_bench_default.json")):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))

# counters / traffic of the PLJ tile kernel, in the form bench.py reads
K = "azp::pair_forces_tiled_kernel<EvalPLJ>"
sq, n = counters("bench_sq", "pair_forces_tiled_kernel<azp::EvalPLJ")
if sq:
    d = {"_comment": "SQ counters per launch of the tile kernel, mean over the launches of `python3 bench.py --no-cpu-baseline --no-side-figures %s` "
                     "(rocprofv3 --pmc, one pass), final binary of round 3" % DRV,
         K: dict(workload="NS", launches=n, **sq)}
    json.dump(d, open(os.path.join(P, "r03_counters.json"), "w"), indent=1)
fe, nf = counters("bench_fetch", "pair_forces_tiled_kernel<azp::EvalPLJ")
wr, nw = counters("bench_write", "pair_forces_tiled_kernel<azp::EvalPLJ")
if fe and wr:
    d = {"_comment": "HBM traffic per launch of the tile kernel, mean over the launches of `python3 bench.py --no-cpu-baseline --no-side-figures %s` (recorded "
                     "cycle + run-in + warm-up + timed + verification launches), separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes. Units KiB; on gfx950 "
                     "FETCH_SIZE reports half the bytes of wide coalesced streams (MI355X_MICROARCH.md, HBM section), hence fetch_correction = 2. The force array is 32 MiB "
                     "(32,768 KiB): WRITE_SIZE above that is register-spill traffic that reached HBM (the kernel keeps 116-224 B of scratch per lane around its tail path; "
                     "41.5 MiB in an earlier pass of the round, none in this one: every full launch wrote exactly 32,768 KiB, the mean includes the early exits)." % DRV,
         K: dict(workload="NS", FETCH_SIZE_KiB=fe["FETCH_SIZE"], WRITE_SIZE_KiB=wr["WRITE_SIZE"], fetch_correction=2.0, launches=nf,
                 note="mean over an MD rebuild cycle at the sustained clock; final binary of round 3")}
    json.dump(d, open(os.path.join(P, "r03_traffic.json"), "w"), indent=1)

# DPD / TwoPatchMorse tile kernels
x = ["# Tile-staged DPD thermostat (C4) and TwoPatchMorse (C5) kernels, round 3: PMC passes", "",
     "`rocprofv3 --kernel-trace --pmc ... -- python3 tools/xtiled_probe.py c4|c5 --reps 5 --settle-ms 0`, mean per launch of `xtiled_kernel<XDPD / XTPM, ...>`", ""]
for name, passes, sub in (("C4 DPD (N = 2,097,152)", ("c4_sq", "c4_fetch", "c4_write"), "xtiled_kernel<azp::XDPD"), ("C5 TwoPatchMorse (N = 524,288)", ("c5_sq",), "xtiled_kernel<azp::XTPM")):
    x += ["## " + name, "", "| counter | mean per launch |", "|---|---|"]
    for p in passes:
        c, n = counters(p, sub)
        for k, v in sorted(c.items()):
            x.append("| %s | %.5g (%d launches) |" % (k, v, n))
    x.append("")
for f, label in (("c4_noprof.log", "C4, HIP events, 80 ms run-in"), ("c5_noprof.log", "C5, HIP events, 80 ms run-in")):
    if os.path.exists(os.path.join(O, f)):
        x.append("%s: `%s`" % (label, open(os.path.join(O, f)).read().strip()[:160]))
open(os.path.join(P, "r03_xtiled_counters.md"), "w").write("\n".join(x) + "\n")

# evaluators
e = ["# The isotropic evaluators on the north-star geometry (N = 1,048,576, <n> = 136.3), round 3", "",
     "`python3 tools/evaluator_probe.py` (HIP events around 30 launches after an 80 ms run-in; fraction = SURVEY 8d bytes / time / 8 TB/s):", "", "```"]
if os.path.exists(os.path.join(O, "evaluators_noprof.log")):
    e += [l.rstrip() for l in open(os.path.join(O, "evaluators_noprof.log"))]
e += ["```", "", "SQ_INSTS_VALU per launch (`rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES -- python3 tools/evaluator_probe.py --reps 5 --settle-ms 0`; the single-type tile "
      "kernel instance of an evaluator serves every single-type row of it, so Colloid SS and CC share one line -- the two-type instance is the `colloid_mix` row):", "",
      "| kernel | launches | SQ_INSTS_VALU |", "|---|---|---|"]
f = newest(os.path.join(O, "eval_sq", "*", "*counter_collection.csv"))
if f:
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_INSTS_VALU" and ("pair_forces" in r["Kernel_Name"]):
            agg[(r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])] += float(r["Counter_Value"])
    per = collections.defaultdict(list)
    for (k, _), v in agg.items():
        per[k].append(v)
    for k, v in sorted(per.items()):
        e.append("| `%s` | %d | %.4g |" % (k.replace("void ", ""), len(v), sum(v) / len(v)))
open(os.path.join(P, "r03_evaluators.md"), "w").write("\n".join(e) + "\n")

# MD logs, A/B, entry
logs = []
for f, label in (("md_bench.log", "north star, under rocprofv3 --kernel-trace --stats"), ("md_bench_noprof.log", "north star, without the profiler"),
                 ("md_bench_buffer07.log", "north star, --buffer 0.7"), ("c3_md.log", "C3 (PerturbedLJ + DoubleWell), under rocprofv3"),
                 ("c3_md_noprof.log", "C3, without the profiler"), ("plan_cells.log", "plan_cells_kernel alone (tools/plan_cells_probe.py: lattice / after --melt 100)")):
    fn = os.path.join(O, f)
    if os.path.exists(fn):
        lines = [l.rstrip() for l in open(fn) if l.strip() and "amdgpu.ids" not in l and "rocprofv3" not in l]
        logs += ["# " + label] + lines[-4:] + [""]
open(os.path.join(P, "r03_md_bench.log"), "w").write("\n".join(logs) + "\n")
ab = ["# A/B of the two tile-kernel options of round 3 (tools/ab_cycle.py: one process, variants alternate pass by pass, a pass = 8 cycle states x 100 launches)", "", "```"]
if os.path.exists(os.path.join(O, "ab_cycle.log")):
    ab += [l.rstrip() for l in open(os.path.join(O, "ab_cycle.log"))]
ab += ["```", "", "phases = the test-free / core-test-free row phases (azp_tuning_set AZP_TUNE_ROW_PHASES), local = per-particle displacement bounds "
       "(azp_pair_args.d_displacement). Both are exact; both issue fewer LDS gathers or VALU instructions and both run SLOWER: off by default."]
if os.path.exists(os.path.join(O, "entry.log")):
    ab += ["", "# The HOOMD-signature entry (tests/test_gpu_auto_plan.py::test_hoomd_signature_entry_runs_at_plan_speed, N = 2^20)", "", "```", open(os.path.join(O, "entry.log")).read().strip(), "```"]
for f in ("c4_noprof.log", "c5_noprof.log"):
    fn = os.path.join(O, f.replace("_noprof", "_entry"))
    if os.path.exists(fn):
        ab += ["", "```", open(fn).read().strip(), "```"]
open(os.path.join(P, "r03_ab_and_entry.md"), "w").write("\n".join(ab) + "\n")
print("\n".join(md[:40]))
