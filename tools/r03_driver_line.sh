# Round 3, item 1: the driver's own command, plain and under rocprofv3 --kernel-trace --stats.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03a}
mkdir -p $O
export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 20 --warmup 5 --settle-ms 0 > $O/bench_driver.json 2> $O/bench_driver.err
python3 -c "
import json,sys
d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1])
print('driver line: ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])
print('by step', [round(x['kernel_ms'],4) for x in d['config']['kernel_ms_by_cycle_step']])
print(d['config']['kernel_ms_other_states'])
"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver2.json 2> $O/bench_driver2.err
python3 -c "
import json,sys
d=json.loads(open('$O/bench_driver2.json').read().strip().splitlines()[-1])
print('driver line (no cpu): ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])
"
rocprofv3 --kernel-trace --stats -d $O/driver_stats --output-format csv -- python3 bench.py --no-cpu-baseline --no-side-figures --steps 20 --warmup 5 > $O/driver_stats.json 2> $O/driver_stats.err
echo done
