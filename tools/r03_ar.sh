cd $GRAFT_REPO_ROOT
python3 tools/xtiled_probe.py c4 2>&1 | grep -v amdgpu | tail -2 | cut -c1-220
python3 tools/xtiled_probe.py c5 2>&1 | grep -v amdgpu | tail -2 | cut -c1-220
