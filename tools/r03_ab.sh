cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "bond or sorter or smoke or full_size or domain or exclusion" > $O/tests.log 2>&1; echo "tests rc $?"; tail -3 $O/tests.log
python3 tools/md_bench.py --workload c3 --steps 300 --dt 0.002 2>&1 | grep -v amdgpu | head -2
