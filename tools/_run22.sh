set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_plan.py tests/test_gpu_full_size.py tests/test_gpu_domain.py -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
rocprofv3 --kernel-trace --stats -d $O/c4_stats --output-format csv -- python3 tools/xtiled_probe.py c4 > $O/c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c5_stats --output-format csv -- python3 tools/xtiled_probe.py c5 > $O/c5.log 2>&1
tail -1 $O/c4.log | cut -c1-400; tail -1 $O/c5.log | cut -c1-400
python3 - <<'PY'
import csv, glob
for d in ("c4_stats","c5_stats"):
    f=glob.glob("gpurun_out/r02t/%s/*/*_kernel_stats.csv"%d)[0]
    for r in list(csv.DictReader(open(f)))[:3]:
        print(d, r["Name"][:70], r["Calls"], "%.1f us"%(float(r["AverageNs"])/1e3))
PY
echo done
