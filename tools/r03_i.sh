set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03i}
mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || (tail -40 $O/tests.log; exit 1)
tail -3 $O/tests.log
for lb in 1 0; do
AZP_LOCAL_BOUND=$lb python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_lb$lb.json 2> $O/bench_lb$lb.err
python3 tools/show_bench.py $O/bench_lb$lb.json
done
