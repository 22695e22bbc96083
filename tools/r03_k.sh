set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k
mkdir -p $O
export TMPDIR=/tmp
python3 tools/ab_cycle.py --passes 5 --per-state 100 local=1 local=0
B="python3 bench.py --no-cpu-baseline --no-side-figures --no-verify --settle-ms 0 --steps 80 --warmup 8"
for lb in 1 0; do
AZP_LOCAL_BOUND=$lb rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/lb$lb --output-format csv -- $B > $O/lb$lb.json 2> $O/lb$lb.err
done
python3 tools/summarize_prof.py $O $O/pmc | grep -i "tiled" 
