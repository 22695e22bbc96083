set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_plan.py tests/test_gpu_full_size.py -x -q 2>&1 | tail -2
AZP_PLAN_CELLS_STOP=0 timeout -k 10 200 python3 tools/plan_cells_probe.py 2>&1 | tail -1 | cut -c1-60
echo done
