set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
export TMPDIR=/tmp
python3 -m pytest tests -x -q -m gpu > gpurun_out/r02g/tests.log 2>&1 || { tail -60 gpurun_out/r02g/tests.log; exit 1; }
tail -3 gpurun_out/r02g/tests.log
python3 tools/config_report.py > gpurun_out/r02g/config_report.md 2> gpurun_out/r02g/config_report.err || { tail -20 gpurun_out/r02g/config_report.err; exit 1; }
cat gpurun_out/r02g/config_report.md
python3 tools/xtiled_probe.py c5 2>&1 | tail -1
echo done
