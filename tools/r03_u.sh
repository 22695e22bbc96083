cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_fused_plan.py tests/test_gpu_full_size.py -x -q > $O/fused.log 2>&1; echo "fused rc $?"; tail -4 $O/fused.log
python3 tools/plan_cells_probe.py 2>&1 | tail -1 | cut -c1-200
python3 tools/plan_cells_probe.py --melt 100 2>&1 | tail -1 | cut -c1-200
python3 tools/md_bench.py --steps 300 2>&1 | grep -v amdgpu | head -3
