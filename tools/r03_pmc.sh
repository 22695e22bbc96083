# PMC pass of the PLJ tile kernel over the bench cycle: tools/r03_pmc.sh <outdir>
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03e}
mkdir -p $O
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-side-figures --no-verify --settle-ms 0 --steps 80 --warmup 8"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/sq --output-format csv -- $B > $O/sq.json 2> $O/sq.err
python3 tools/summarize_prof.py $O $O/pmc | grep -i "tiled" 
