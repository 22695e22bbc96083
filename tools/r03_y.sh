cd $GRAFT_REPO_ROOT
for b in 0.4 0.5 0.6 0.7; do echo "buffer $b"; python3 tools/md_bench.py --steps 300 --buffer $b 2>&1 | grep -v amdgpu | head -1 | cut -c1-200; done
