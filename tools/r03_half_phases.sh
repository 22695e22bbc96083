#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r03half
O=gpurun_out/r03half/phases.log
: > $O
for h in 0 1; do
  for st in 1 2 258 514 770 4 0; do
    echo "== half=$h stop=$st" >> $O
    AZP_HALF_CELLS=$h AZP_LIB_PATH=tools/libazp_pcprof.so AZP_PLAN_CELLS_STOP=$st timeout -k 10 200 python tools/plan_cells_probe.py --melt 100 >> $O 2>&1 || exit 1
  done
done
grep -E "==|build_from" $O | sed 's/info.*//'
# counters of the whole kernel in both forms (production library)
export TMPDIR=/tmp
for h in 0 1; do
  P=gpurun_out/r03half/pmc_$h
  AZP_HALF_CELLS=$h rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $P --output-format csv -- python3 tools/plan_cells_probe.py --melt 100 > $P.log 2>&1 || exit 1
  python3 tools/summarize_prof.py $P $P/pmc | grep -i "plan_cells" || true
done
