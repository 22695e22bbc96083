cd $GRAFT_REPO_ROOT
O=gpurun_out/r03s
mkdir -p $O
export TMPDIR=/tmp
python3 tools/md_bench.py --steps 300 2>&1 | grep -v amdgpu | tee $O/md_noprof.log
rocprofv3 --kernel-trace --stats -d $O/md_stats --output-format csv -- python3 tools/md_bench.py --steps 300 > $O/md_prof.log 2>&1
tail -4 $O/md_prof.log
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r03s/md_stats/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-70s calls %6s total %9.3f ms avg %9.1f us  %s%%"%(r['Name'][:70],r['Calls'],float(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e3,r['Percentage'][:5]))
PY
