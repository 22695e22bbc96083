cd $GRAFT_REPO_ROOT
for k in 1 2 3; do python3 tools/md_bench.py --workload c3 --steps 300 --dt 0.002 2>&1 | grep -v amdgpu | head -1 | cut -c1-90; done
