cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m
mkdir -p $O
timeout -k 10 500 python3 -m pytest tests/test_gpu_auto_plan.py -x -q -s > $O/auto.log 2>&1; echo "auto rc $?"; tail -25 $O/auto.log
