set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_plan.py -x -q 2>&1 | tail -2
for s in 0 2 514 4; do AZP_PLAN_CELLS_STOP=$s timeout -k 10 200 python3 tools/plan_cells_probe.py 2>&1 | tail -1 | cut -c1-60; done
echo done
