cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "bond or external or nve or smoke or full_size or reference" > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
python3 tools/md_bench.py --workload c3 --steps 300 --dt 0.002 2>&1 | grep -v amdgpu | head -2
rocprofv3 --kernel-trace --stats -d $O/c3_stats --output-format csv -- python3 tools/md_bench.py --workload c3 --steps 300 --dt 0.002 > $O/c3_prof.log 2>&1
grep "ms/step" $O/c3_prof.log
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r03z/c3_stats/*/*_kernel_stats.csv')[0]
tot=0
for r in list(csv.DictReader(open(f)))[:12]:
    print("%-66s calls %6s total %9.3f ms avg %9.1f us  %s%%"%(r['Name'][:66],r['Calls'],float(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e3,r['Percentage'][:5]))
print("sum of all kernels: %.3f ms"%(sum(float(r['TotalDurationNs']) for r in csv.DictReader(open(f)))/1e6))
PY
