set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03d}
mkdir -p $O
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || (tail -30 $O/tests.log; exit 1)
tail -3 $O/tests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
python3 tools/show_bench.py $O/bench_driver.json
