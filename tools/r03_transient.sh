# clock transient of the tile kernel: 600 back-to-back launches on one state, per-launch durations from the kernel trace
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03b}
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/static_trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-side-figures --static --steps 600 --warmup 0 --settle-ms 0 > $O/static.json 2> $O/static.err
rocprofv3 --kernel-trace -d $O/cycle_trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-side-figures --steps 400 --warmup 0 --settle-ms 0 > $O/cycle.json 2> $O/cycle.err
echo done
