set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03g}
mkdir -p $O
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-side-figures --no-verify --settle-ms 0 --steps 80 --warmup 8"
for ph in 1 0; do
export AZP_ROW_PHASES=$ph
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/sq$ph --output-format csv -- $B > $O/sq$ph.json 2> $O/sq$ph.err
python3 bench.py --no-cpu-baseline --steps 200 --warmup 20 > $O/bench$ph.json 2> $O/bench$ph.err
python3 tools/show_bench.py $O/bench$ph.json
done
python3 tools/summarize_prof.py $O $O/pmc | grep -i "tiled" 
