# The rocprofv3 passes behind profiles/r03_*: run on a GPU box from the repository root
#   gpurun -- bash tools/profile_passes_r03.sh [part]      part = a (bench) | b (md, configs) | c (evaluators, entry)
# then python3 tools/condense_profiles_r03.py gpurun_out/r03final writes the summaries under profiles/.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final
mkdir -p $O
export TMPDIR=/tmp
PART=${1:-abc}
DRV="--steps 20 --warmup 5"          # the driver's arguments
B="python3 bench.py --no-cpu-baseline --no-side-figures $DRV"
if [[ $PART == *a* ]]; then
python3 bench.py --gpus 1 $DRV > $O/bench_driver.json 2> $O/bench_driver.err
python3 tools/show_bench.py $O/bench_driver.json
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 tools/show_bench.py $O/bench_default.json
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- $B > $O/bench_stats.json 2> $O/bench_stats.err
echo "bench stats pass done"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/bench_sq --output-format csv -- $B > $O/bench_sq.json 2> $O/bench_sq.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/bench_fetch --output-format csv -- $B > $O/bench_fetch.json 2> $O/bench_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/bench_write --output-format csv -- $B > $O/bench_write.json 2> $O/bench_write.err
python3 tools/ab_cycle.py --passes 4 --per-state 100 phases=0,local=0 phases=1,local=0 phases=0,local=1 2>&1 | grep -v amdgpu > $O/ab_cycle.log
cat $O/ab_cycle.log
fi
if [[ $PART == *b* ]]; then
rocprofv3 --kernel-trace --stats -d $O/md_stats --output-format csv -- python3 tools/md_bench.py --steps 300 > $O/md_bench.log 2> $O/md_bench.err
tail -4 $O/md_bench.log
python3 tools/md_bench.py --steps 300 > $O/md_bench_noprof.log 2>&1
tail -4 $O/md_bench_noprof.log
python3 tools/md_bench.py --steps 300 --buffer 0.7 > $O/md_bench_buffer07.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c3_stats --output-format csv -- python3 tools/md_bench.py --workload c3 --steps 300 > $O/c3_md.log 2> $O/c3_md.err
python3 tools/md_bench.py --workload c3 --steps 300 > $O/c3_md_noprof.log 2>&1
tail -4 $O/c3_md_noprof.log
rocprofv3 --kernel-trace --stats -d $O/c4_stats --output-format csv -- python3 tools/xtiled_probe.py c4 > $O/c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c5_stats --output-format csv -- python3 tools/xtiled_probe.py c5 > $O/c5.log 2>&1
tail -1 $O/c4.log | cut -c1-300; tail -1 $O/c5.log | cut -c1-300
python3 tools/xtiled_probe.py c4 2>&1 | tail -1 | cut -c1-200 > $O/c4_noprof.log; cat $O/c4_noprof.log
python3 tools/xtiled_probe.py c5 2>&1 | tail -1 | cut -c1-200 > $O/c5_noprof.log; cat $O/c5_noprof.log
rocprofv3 --kernel-trace --stats -d $O/bond_stats --output-format csv -- python3 tools/bond_probe.py > $O/bond.log 2>&1
tail -1 $O/bond.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/c4_sq --output-format csv -- python3 tools/xtiled_probe.py c4 --reps 5 --settle-ms 0 > $O/c4_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/c4_fetch --output-format csv -- python3 tools/xtiled_probe.py c4 --reps 5 --settle-ms 0 > $O/c4_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/c4_write --output-format csv -- python3 tools/xtiled_probe.py c4 --reps 5 --settle-ms 0 > $O/c4_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES -d $O/c5_sq --output-format csv -- python3 tools/xtiled_probe.py c5 --reps 5 --settle-ms 0 > $O/c5_sq.log 2>&1
python3 tools/plan_cells_probe.py 2>&1 | tail -1 | cut -c1-60 > $O/plan_cells.log
python3 tools/plan_cells_probe.py --melt 100 2>&1 | tail -1 | cut -c1-60 >> $O/plan_cells.log
cat $O/plan_cells.log
fi
if [[ $PART == *c* ]]; then
rocprofv3 --kernel-trace --stats -d $O/eval_stats --output-format csv -- python3 tools/evaluator_probe.py > $O/evaluators.log 2> $O/evaluators.err
grep -v amdgpu $O/evaluators.log | grep "N="
python3 tools/evaluator_probe.py 2>&1 | grep "N=" > $O/evaluators_noprof.log
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES -d $O/eval_sq --output-format csv -- python3 tools/evaluator_probe.py --reps 5 --settle-ms 0 > $O/evaluators_sq.log 2>&1
python3 -m pytest tests/test_gpu_auto_plan.py -q -s -k plan_speed 2>&1 | grep "HOOMD-signature" > $O/entry.log
cat $O/entry.log
rocprofv3 --kernel-trace --stats -d $O/entry_stats --output-format csv -- python3 -m pytest tests/test_gpu_auto_plan.py -q -s -k plan_speed > $O/entry_prof.log 2>&1
fi
python3 tools/summarize_prof.py $O $O/all > /dev/null 2>&1 || true
echo done
