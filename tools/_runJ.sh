cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02y
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r02y/gpu_tests.log 2>&1
echo "pytest rc $?"
tail -3 gpurun_out/r02y/gpu_tests.log | cut -c1-200
python3 tools/md_bench.py --steps 300 2>&1 | tail -4 | head -1 | cut -c1-100
rocprofv3 --kernel-trace --stats -d gpurun_out/r02y/md_stats --output-format csv -- python3 tools/md_bench.py --steps 300 > gpurun_out/r02y/md_bench.log 2> gpurun_out/r02y/md_bench.err
python3 - <<'PY'
import csv, glob
f=glob.glob("gpurun_out/r02y/md_stats/*/*_kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print(r["Name"][:60], r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3), "%.1f ms"%(float(r["TotalDurationNs"])/1e6))
PY
echo done
