cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ah
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -4 $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; python3 tools/show_bench.py $O/bench.json | head -2
