set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 400 python3 tools/md_bench.py --steps 300 > $O/md_bench_noprof.log 2>&1
tail -3 $O/md_bench_noprof.log
timeout -k 10 200 python3 tools/plan_cells_probe.py --melt 100 2>&1 | tail -1 | cut -c1-60
echo done
