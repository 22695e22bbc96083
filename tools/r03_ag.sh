cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final; mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/c4_stats $O/c5_stats $O/eval_stats
rocprofv3 --kernel-trace --stats -d $O/c4_stats --output-format csv -- python3 tools/xtiled_probe.py c4 > $O/c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c5_stats --output-format csv -- python3 tools/xtiled_probe.py c5 > $O/c5.log 2>&1
tail -1 $O/c4.log | cut -c1-200; tail -1 $O/c5.log | cut -c1-200
python3 tools/xtiled_probe.py c4 2>&1 | tail -1 | cut -c1-200 > $O/c4_noprof.log; cat $O/c4_noprof.log
python3 tools/xtiled_probe.py c5 2>&1 | tail -1 | cut -c1-200 > $O/c5_noprof.log; cat $O/c5_noprof.log
rocprofv3 --kernel-trace --stats -d $O/eval_stats --output-format csv -- python3 tools/evaluator_probe.py > $O/evaluators.log 2> $O/evaluators.err
python3 tools/evaluator_probe.py 2>&1 | grep "N=" > $O/evaluators_noprof.log
cat $O/evaluators_noprof.log
rocprofv3 --kernel-trace --stats -d $O/bond_stats --output-format csv -- python3 tools/bond_probe.py > $O/bond.log 2>&1
tail -2 $O/bond.log
