cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do AZP_NATIVE_BINNING=$v python3 tools/md_bench.py --steps 300 2>&1 | grep -v amdgpu | head -1 | cut -c1-80; done
