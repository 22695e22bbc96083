"""Per-launch durations (us) of the last N kernels of a rocprofv3 kernel trace, in start order;
a gap of more than 1 us before a launch is shown as (gNN)."""
import csv, glob, sys
d, n = sys.argv[1], int(sys.argv[2])
pat = sys.argv[3] if len(sys.argv) > 3 else ""
f = glob.glob(d + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
prev = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    g = (s - prev) / 1e3 if prev else 0.0
    prev = e
    if pat and pat not in r["Kernel_Name"]:
        continue
    out.append("%.0f%s" % ((e - s) / 1e3, "" if g < 1 else "(g%.0f)" % g))
print(" ".join(out[-n:]))
