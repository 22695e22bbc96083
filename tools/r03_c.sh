set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03c}
mkdir -p $O
export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
python3 tools/show_bench.py $O/bench_driver.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --balance 1 > $O/bench_bal.json 2> $O/bench_bal.err
python3 tools/show_bench.py $O/bench_bal.json
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python3 tools/show_bench.py $O/bench_default.json
