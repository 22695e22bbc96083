set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
export TMPDIR=/tmp
python3 -m pytest tests/test_gpu_domain.py tests/test_gpu_auto_plan.py -x -q -m gpu > gpurun_out/r02d/tests.log 2>&1 || { tail -60 gpurun_out/r02d/tests.log; exit 1; }
tail -3 gpurun_out/r02d/tests.log
python3 bench.py --no-cpu-baseline > gpurun_out/r02d/bench_resident.json 2> gpurun_out/r02d/bench.err || { tail -20 gpurun_out/r02d/bench.err; exit 1; }
python3 bench.py --no-cpu-baseline --stage-positions 1 > gpurun_out/r02d/bench_staged.json 2> gpurun_out/r02d/bench2.err || { tail -20 gpurun_out/r02d/bench2.err; exit 1; }
python3 - <<'PY'
import json
for f in ("resident","staged"):
    d=json.load(open("gpurun_out/r02d/bench_%s.json"%f))
    print(f, "kernel_ms %.4f frac %.3f"%(d["roofline"]["kernel_ms"], d["roofline"]["frac"]), "by-step mean %.4f"%(sum(r["kernel_ms"] for r in d["config"]["kernel_ms_by_cycle_step"])/len(d["config"]["kernel_ms_by_cycle_step"])))
PY
for v in main soapad; do
  if [ $v = main ]; then unset AZP_LIB_PATH; else export AZP_LIB_PATH=$GRAFT_REPO_ROOT/tools/libazp_$v.so; fi
  python3 tools/cycle_probe.py --bank 0 > gpurun_out/r02d/probe_$v.log 2>&1
  grep mean_ms gpurun_out/r02d/probe_$v.log | cut -c1-220
done
unset AZP_LIB_PATH
AZP_BENCH_FORCE_DD=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 AZP_DIST_BACKEND=gloo python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r02d/dd1.json 2> gpurun_out/r02d/dd1.err || { tail -20 gpurun_out/r02d/dd1.err; exit 1; }
cut -c1-300 gpurun_out/r02d/dd1.json
for w in c4 c5 ns; do
AZP_DIST_BACKEND=gloo AZP_BENCH_ONE_DEVICE=1 AZP_BENCH_VERIFY=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 10 --warmup 2 --workload $w --no-cpu-baseline > gpurun_out/r02d/dd2_$w.json 2> gpurun_out/r02d/dd2_$w.err || { tail -30 gpurun_out/r02d/dd2_$w.err; exit 1; }
python3 - $w <<'PY'
import json,sys
d=json.load(open("gpurun_out/r02d/dd2_%s.json"%sys.argv[1]))
print(sys.argv[1], "value %.3e ms/step %.3f verify %r kernel_ms %.3f"%(d["value"], d["ms_per_step"], d["config"]["max_rel_error_vs_single_domain"], d["roofline"]["kernel_ms"]))
PY
done
echo done
