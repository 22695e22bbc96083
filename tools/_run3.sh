set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
export TMPDIR=/tmp
python3 -m pytest tests -x -q -m gpu > gpurun_out/r02c/tests.log 2>&1 || { tail -40 gpurun_out/r02c/tests.log; exit 1; }
tail -3 gpurun_out/r02c/tests.log
python3 bench.py > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err || { tail -20 gpurun_out/r02c/bench.err; exit 1; }
cut -c1-600 gpurun_out/r02c/bench.json
for v in main soapad; do
  if [ $v = main ]; then unset AZP_LIB_PATH; else export AZP_LIB_PATH=$GRAFT_REPO_ROOT/tools/libazp_$v.so; fi
  python3 tools/cycle_probe.py --bank 0 > gpurun_out/r02c/probe_$v.log 2>&1
  grep mean_ms gpurun_out/r02c/probe_$v.log | cut -c1-220
done
echo done
