set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
export TMPDIR=/tmp
python3 -m pytest tests -x -q -m gpu > gpurun_out/r02e/tests.log 2>&1 || { tail -60 gpurun_out/r02e/tests.log; exit 1; }
tail -3 gpurun_out/r02e/tests.log
python3 tools/config_report.py > gpurun_out/r02e/config_report.md 2> gpurun_out/r02e/config_report.err || { tail -20 gpurun_out/r02e/config_report.err; exit 1; }
cat gpurun_out/r02e/config_report.md
python3 bench.py > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err || { tail -20 gpurun_out/r02e/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r02e/bench.json"))
print("bench kernel_ms %.4f frac %.3f"%(d["roofline"]["kernel_ms"], d["roofline"]["frac"]), "by-step mean %.4f"%(sum(r["kernel_ms"] for r in d["config"]["kernel_ms_by_cycle_step"])/len(d["config"]["kernel_ms_by_cycle_step"])), d["config"]["kernel_ms_other_states"])
PY
echo done
