set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "planned_dpd or planned_aniso" > gpurun_out/r02f/tests.log 2>&1 || { tail -40 gpurun_out/r02f/tests.log; exit 1; }
tail -2 gpurun_out/r02f/tests.log
python3 tools/xtiled_probe.py c4 2>&1 | tail -1
python3 tools/xtiled_probe.py c5 2>&1 | tail -1
AZP_LIB_PATH=$GRAFT_REPO_ROOT/tools/libazp_xdpd_stage.so python3 tools/xtiled_probe.py c4 2>&1 | tail -1
AZP_LIB_PATH=$GRAFT_REPO_ROOT/tools/libazp_xtpm_stage.so python3 tools/xtiled_probe.py c5 2>&1 | tail -1
echo done
