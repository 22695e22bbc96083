set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/pc_sq --output-format csv -- python3 tools/plan_cells_probe.py --reps 3 > $O/pc_sq.log 2> $O/pc_sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM -d $O/pc_sq2 --output-format csv -- python3 tools/plan_cells_probe.py --reps 3 > $O/pc_sq2.log 2> $O/pc_sq2.err
tail -1 $O/pc_sq.log | cut -c1-100
python3 - <<'PY'
import csv, glob, collections
for d in ("gpurun_out/r02o/pc_sq", "gpurun_out/r02o/pc_sq2"):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f:
        print("no counter file in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        if "plan_cells" in k:
            print(k, {a: "%.3g" % b for a, b in v.items()})
PY
echo done
