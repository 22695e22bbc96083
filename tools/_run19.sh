set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests/test_gpu_fused_plan.py tests/test_gpu_domain.py -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
AZP_BENCH_FORCE_DD=1 AZP_BENCH_FORCE_OVERLAP=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 3 2>$O/dd1.err | tail -1 | cut -c1-700
echo done
