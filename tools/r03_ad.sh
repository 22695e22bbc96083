cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ad
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
python3 tools/md_bench.py --steps 300 2>&1 | grep -v amdgpu | head -1
python3 tools/md_bench.py --workload c3 --steps 300 --dt 0.002 2>&1 | grep -v amdgpu | head -1
