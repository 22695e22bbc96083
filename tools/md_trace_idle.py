"""GPU busy / idle of the timed steps of a `rocprofv3 --kernel-trace -- python3 tools/md_bench.py ...` run:
    python tools/md_trace_idle.py <dir with */*_kernel_trace.csv>
Per step: kernel time by kernel, idle time by the kernel that follows the gap."""
import collections
import csv
import glob
import os
import sys

f = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
nve = [i for i, r in enumerate(rows) if "nve_kernel" in r["Kernel_Name"] or "nve_rot" in r["Kernel_Name"]]
k = len(nve) - 1
while k > 0 and int(rows[nve[k]]["Start_Timestamp"]) - int(rows[nve[k - 1]]["End_Timestamp"]) < 20e6:
    k -= 1
rows = rows[nve[k]:nve[-1] + 1]
steps = len(nve) - k
t0, t1 = int(rows[0]["Start_Timestamp"]), int(rows[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
print("%d steps: %.4f ms per step, GPU busy %.1f %%, idle %.4f ms per step" % (steps, (t1 - t0) / 1e6 / steps, 100.0 * busy / (t1 - t0), (t1 - t0 - busy) / 1e6 / steps))
gaps, cnt, ktime, kc = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"][:48]
    ktime[name] += e - s
    kc[name] += 1
    if prev is not None and s > prev:
        gaps[name] += s - prev
        cnt[name] += 1
    prev = max(prev or 0, e)
print("kernel time, ms per step")
for name, v in ktime.most_common(8):
    print("  %8.4f  %5d x %s" % (v / 1e6 / steps, kc[name], name))
print("idle before, ms per step")
for name, v in gaps.most_common(6):
    print("  %8.4f  %5d x %s" % (v / 1e6 / steps, cnt[name], name))
