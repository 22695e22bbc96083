#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03half
O=gpurun_out/r03half/count.log
: > $O
for h in 0 1; do
  echo "== half=$h" >> $O
  AZP_HALF_CELLS=$h AZP_LIB_PATH=tools/libazp_pcprof.so timeout -k 10 200 python tools/plan_cells_probe.py --melt 100 --reps 2 2>&1 | grep -E "plan_cells:|build_from" | tail -3 | sed 's/info.*//' >> $O || exit 1
done
cat $O
