set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03f}
mkdir -p $O
export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver.json 2> $O/bench_driver.err
python3 tools/show_bench.py $O/bench_driver.json
python3 -c "
import json
d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
print(d['config']['tile_plan'])
"
