#!/bin/bash
# half-width cells in the plan compiler: tests, then timings
set -o pipefail
mkdir -p gpurun_out/r03half
O=gpurun_out/r03half
timeout -k 10 900 python -m pytest tests/test_gpu_fused_plan.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for h in 0 1; do
  echo "== AZP_HALF_CELLS=$h" | tee -a $O/probe.log
  AZP_HALF_CELLS=$h timeout -k 10 300 python tools/plan_cells_probe.py >> $O/probe.log 2>&1 || exit 1
  AZP_HALF_CELLS=$h timeout -k 10 300 python tools/plan_cells_probe.py --melt 100 >> $O/probe.log 2>&1 || exit 1
  AZP_HALF_CELLS=$h timeout -k 10 300 python tools/md_bench.py --steps 300 >> $O/md_$h.log 2>&1 || exit 1
  tail -4 $O/md_$h.log
done
grep -E "==|build_from_cells" $O/probe.log
