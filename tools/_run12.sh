set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_plan.py -x -q > $O/fused_tests.log 2>&1 || { tail -60 $O/fused_tests.log; exit 1; }
tail -3 $O/fused_tests.log
for s in 0 1 2 4; do AZP_PLAN_CELLS_STOP=$s timeout -k 10 200 python3 tools/plan_cells_probe.py 2>&1 | tail -1; done
timeout -k 10 200 python3 tools/plan_cells_probe.py --melt 100 2>&1 | tail -1
echo done
