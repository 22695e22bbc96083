cd $GRAFT_REPO_ROOT
AZP_DIST_BACKEND=gloo AZP_BENCH_FORCE_DD=1 AZP_BENCH_FORCE_OVERLAP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 RANK=0 WORLD_SIZE=1 python3 bench.py --gpus 1 --workload ns-small --steps 200 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('host-bound step (16k particles, two launches + pack + collective of one rank): %.4f ms per step; kernel_ms %.4f'%(d['ms_per_step'], d['roofline']['kernel_ms']))
"
python3 tools/dd_plan_check.py --strong --worlds 2,4,8 2>&1 | grep world
python3 tools/dd_plan_check.py --strong --worlds 4,8 --tpp 2 2>&1 | grep world
python3 tools/dd_plan_check.py --strong --worlds 8 --tpp 4 2>&1 | grep world
