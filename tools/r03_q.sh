cd $GRAFT_REPO_ROOT
O=gpurun_out/r03q
mkdir -p $O
export TMPDIR=/tmp
for v in new r02; do
  if [ $v = r02 ]; then export AZP_LIB_PATH=tools/libazp_colloid_r02.so; fi
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/cc_$v --output-format csv -- python3 tools/evaluator_probe.py colloid_cc colloid_mix colloid_ss --reps 5 > $O/cc_$v.log 2>&1
  grep -v amdgpu $O/cc_$v.log | grep colloid
done
python3 tools/summarize_prof.py $O $O/pmc | grep -E "tiled|forces_kernel" | grep INSTS
