cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ac
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU -d $O/c4_sq --output-format csv -- python3 tools/xtiled_probe.py c4 --reps 5 > $O/c4_sq.log 2>&1
python3 tools/summarize_prof.py $O $O/pmc | grep -E "xtiled|dpd_forces" 
