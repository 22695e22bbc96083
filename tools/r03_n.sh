cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/auto_stats --output-format csv -- python3 -m pytest tests/test_gpu_auto_plan.py -x -q -s -k plan_speed > $O/auto.log 2>&1
tail -3 $O/auto.log
python3 tools/trace_tail.py $O/auto_stats 70
