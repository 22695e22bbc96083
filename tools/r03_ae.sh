cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x
timeout -k 10 300 python3 -m pytest tests/test_gpu_fused_plan.py -x -q -k "binning or parity or random" 2>&1 | tail -2
bash tools/r03_x.sh
