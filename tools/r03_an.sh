cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
for v in 1 0 1 0; do AZP_SPLIT_TILES=$v python3 tools/md_bench.py --steps 300 2>&1 | grep -v amdgpu | sed -n '1p;3p' | cut -c1-110; done
