set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03h}
mkdir -p $O
AZP_LIB_PATH=tools/libazp_dbg.so python3 bench.py --no-cpu-baseline --no-side-figures --no-verify --settle-ms 0 --steps 8 --warmup 0 > $O/dbg.json 2> $O/dbg.err
grep "tile 7 wave 0" $O/dbg.json | tail -12
