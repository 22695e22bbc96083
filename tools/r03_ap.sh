cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ap; mkdir -p $O; export TMPDIR=/tmp
python3 tools/md_bench.py --workload c5 --steps 300 --dt 0.004 2>&1 | grep -v amdgpu | head -1 | cut -c1-140
python3 tools/md_bench.py --workload c4 --steps 300 --dt 0.01 2>&1 | grep -v amdgpu | head -1 | cut -c1-140
rocprofv3 --kernel-trace --stats -d $O/c5 --output-format csv -- python3 tools/md_bench.py --workload c5 --steps 300 --dt 0.004 > $O/c5.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r03ap/c5/*/*_kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:8]:
    print("%-60s calls %5s total %8.3f ms avg %7.1f us"%(r['Name'][:60],r['Calls'],float(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e3))
print("sum %.2f ms"%(sum(float(r['TotalDurationNs']) for r in rows)/1e6))
PY
grep "ms/step" $O/c5.log | cut -c1-100
