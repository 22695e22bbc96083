cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "parity or reference or golden or hertz or aniso or dpd or colloid or auto" > $O/tests.log 2>&1; echo "tests rc $?"; tail -5 $O/tests.log
python3 tools/evaluator_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/evaluators.log
python3 tools/xtiled_probe.py c4 2>&1 | tail -1 | tee $O/c4.log
python3 tools/xtiled_probe.py c5 2>&1 | tail -1 | tee $O/c5.log
