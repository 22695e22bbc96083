set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
export TMPDIR=/tmp
python3 tools/rot_energy_probe.py 2>&1 | grep "^rot"
python3 -m pytest tests/test_gpu_auto_plan.py -x -q -m gpu -s -k plan_speed 2>&1 | grep -E "HOOMD-signature|passed|failed"
python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_external_nve.py::test_two_patch_morse_nve_with_rotation_conserves_energy > gpurun_out/r02h/tests.log 2>&1 || { tail -60 gpurun_out/r02h/tests.log; exit 1; }
tail -3 gpurun_out/r02h/tests.log
python3 tools/config_report.py > gpurun_out/r02h/config_report.md 2> gpurun_out/r02h/config_report.err || { tail -20 gpurun_out/r02h/config_report.err; exit 1; }
cat gpurun_out/r02h/config_report.md
python3 tools/xtiled_probe.py c5 2>&1 | tail -1
echo done
