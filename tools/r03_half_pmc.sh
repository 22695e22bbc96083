#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/r03half
for h in 0 1; do
  P=gpurun_out/r03half/pmc2_$h
  rm -rf $P
  AZP_HALF_CELLS=$h rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $P --output-format csv -- python3 tools/plan_cells_probe.py --melt 100 > $P.log 2>&1 || exit 1
  P=gpurun_out/r03half/pmc3_$h
  rm -rf $P
  AZP_HALF_CELLS=$h rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_ANY -d $P --output-format csv -- python3 tools/plan_cells_probe.py --melt 100 > $P.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv,glob,collections
for h in (0,1):
  for pp in ('pmc2','pmc3'):
    fs=glob.glob('gpurun_out/r03half/%s_%d/*/*_counter_collection.csv'%(pp,h))
    if not fs: print('no file',pp,h); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'plan_cells' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print('half=%d'%h, '  '.join('%s=%.4g'%(c,sum(v)/len(v)) for c,v in sorted(acc.items())))
PY
