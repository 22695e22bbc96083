"""One-paragraph digest of a bench.py JSON line (file argument)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("%s: steps %d warmup %d  ms_per_step %.4f  kernel_ms %.4f  frac %.3f  value %.3e  verified %s" % (
    sys.argv[1], d["steps"], d["warmup"], d["ms_per_step"], r["kernel_ms"], r["frac"], d["value"], d.get("verified")))
bs = [x["kernel_ms"] for x in d["config"].get("kernel_ms_by_cycle_step", [])]
if bs:
    print("  by cycle step:", " ".join("%.4f" % x for x in bs), " mean %.4f" % (sum(bs) / len(bs)))
for k, v in d["config"].get("kernel_ms_other_states", {}).items():
    print("  %s: %.4f" % (k, v))
if "cpu_baseline" in d:
    c = d["cpu_baseline"]
    print("  cpu: %.3e (%d core)  gpu/cpu %.0f  err vs gpu %s" % (c["value"], c["cores"], c.get("gpu_over_cpu_1core", 0), c.get("gpu_vs_cpu_max_abs_err_over_max_force")))
