#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof/<pass>/...) into the small
summaries kept under profiles/: per-kernel stats and per-kernel mean counters.

    python tools/summarize_prof.py gpurun_out/prof profiles/r01_plj_generic
"""
import collections
import csv
import glob
import os
import sys


def main(src, dst_prefix, match="azp::"):
    out = []
    for f in sorted(glob.glob(os.path.join(src, "*", "*", "*_kernel_stats.csv"))):
        out.append("# kernel stats (%s)" % os.path.relpath(f, src))
        out.append("name,calls,total_ns,avg_ns,pct,min_ns,max_ns,stddev")
        for r in csv.DictReader(open(f)):
            if match in r["Name"]:
                out.append(",".join('"%s"' % r["Name"] if k == "Name" else r[k] for k in
                                    ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")))
    out.append("")
    out.append("# PMC counters: mean per launch, per kernel (separate rocprofv3 --pmc passes)")
    out.append("pass,kernel,counter,launches,mean")
    for f in sorted(glob.glob(os.path.join(src, "*", "*", "*_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        p = os.path.relpath(f, src).split(os.sep)[0]
        for (k, c), v in sorted(agg.items()):
            out.append('%s,"%s",%s,%d,%.6g' % (p, k, c, len(v), sum(v) / len(v)))
    open(dst_prefix + "_summary.csv", "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main(*sys.argv[1:])
