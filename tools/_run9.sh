set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_external_nve.py -x -q -m gpu 2>&1 | tail -3
python3 bench.py > gpurun_out/r02i/bench.json 2> gpurun_out/r02i/bench.err || { tail -20 gpurun_out/r02i/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r02i/bench.json"))
print("bench value %.4g kernel_ms %.4f frac %.3f"%(d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"]), "by-step mean %.4f"%(sum(r["kernel_ms"] for r in d["config"]["kernel_ms_by_cycle_step"])/len(d["config"]["kernel_ms_by_cycle_step"])), d["config"]["kernel_ms_other_states"], d["cpu_baseline"]["value"])
PY
python3 tools/bond_probe.py
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/r02i/bond_fetch --output-format csv -- python3 tools/bond_probe.py 20 > gpurun_out/r02i/bond_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/r02i/bond_write --output-format csv -- python3 tools/bond_probe.py 20 > gpurun_out/r02i/bond_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/r02i/bond_sq --output-format csv -- python3 tools/bond_probe.py 20 > gpurun_out/r02i/bond_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum -d gpurun_out/r02i/bond_tc --output-format csv -- python3 tools/bond_probe.py 20 > gpurun_out/r02i/bond_tc.log 2>&1 || true
python3 tools/summarize_prof.py gpurun_out/r02i gpurun_out/r02i/bond 2>&1 | grep -i "bond" | head -30
echo done
