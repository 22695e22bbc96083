set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests/test_gpu_fused_plan.py -x -q > $O/fused_tests.log 2>&1 || { tail -60 $O/fused_tests.log; exit 1; }
tail -3 $O/fused_tests.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/md_stats --output-format csv -- python3 tools/md_bench.py --steps 300 > $O/md_bench.log 2> $O/md_bench.err || { tail -30 $O/md_bench.err; exit 1; }
tail -4 $O/md_bench.log
timeout -k 10 300 python3 tools/cycle_probe.py --reps 20 --fused 1 > $O/cycle_fused.log 2>&1 || { tail -30 $O/cycle_fused.log; exit 1; }
tail -1 $O/cycle_fused.log | cut -c1-600
timeout -k 10 300 python3 tools/cycle_probe.py --reps 20 --fused 0 > $O/cycle_list.log 2>&1
tail -1 $O/cycle_list.log | cut -c1-600
python3 tools/summarize_prof.py $O $O/all > /dev/null 2>&1 || true
echo done
