# bench.py --gpus N rehearsed on ONE GPU (gloo for the collectives, every rank on device 0), with the decomposed
# forces checked against a single-domain evaluation: tools/rehearse_multi.sh [outdir]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-rehearse}
mkdir -p $O
export AZP_DIST_BACKEND=gloo AZP_BENCH_ONE_DEVICE=1 AZP_BENCH_VERIFY=1
for spec in "2 ns" "4 ns" "2 c4" "4 c5"; do
  set -- $spec
  timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port $((29500 + $1)) bench.py --gpus $1 --steps 10 --warmup 3 --workload $2 > $O/r$1_$2.json 2> $O/r$1_$2.err
  echo "ranks $1 workload $2 rc $?"
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/r$1_$2.json").read().strip().splitlines()[-1])
    print("  value %.3e  ms_per_step %.4f  kernel_ms %.4f  max rel error vs single domain %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["max_rel_error_vs_single_domain"]))
except Exception as e:
    print("  no JSON line:", e)
    print(open("$O/r$1_$2.err").read()[-1500:])
PY
done
