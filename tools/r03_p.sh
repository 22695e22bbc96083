cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p
mkdir -p $O
python3 tools/evaluator_probe.py hertz yukawa dpd_cons colloid_ss colloid_cc colloid_mix 2>&1 | grep -v amdgpu.ids | tee $O/evaluators.log
