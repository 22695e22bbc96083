cd $GRAFT_REPO_ROOT
python3 -m cProfile -s tottime tools/md_bench.py --workload c3 --steps 200 --dt 0.002 2>&1 | grep -v amdgpu | head -40
