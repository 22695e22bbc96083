set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02j
mkdir -p $O
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-side-figures --steps 80 --warmup 8"
rocprofv3 --kernel-trace --stats -d $O/bench_stats --output-format csv -- $B > $O/bench_stats.json 2> $O/bench_stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/bench_fetch --output-format csv -- $B > $O/bench_fetch.json 2> $O/bench_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/bench_write --output-format csv -- $B > $O/bench_write.json 2> $O/bench_write.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d $O/bench_sq --output-format csv -- $B > $O/bench_sq.json 2> $O/bench_sq.err
echo "bench passes done"
rocprofv3 --kernel-trace --stats -d $O/md_stats --output-format csv -- python3 tools/md_bench.py --steps 300 > $O/md_bench.log 2> $O/md_bench.err
tail -4 $O/md_bench.log
rocprofv3 --kernel-trace --stats -d $O/c4_stats --output-format csv -- python3 tools/xtiled_probe.py c4 > $O/c4.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/c5_stats --output-format csv -- python3 tools/xtiled_probe.py c5 > $O/c5.log 2>&1
tail -1 $O/c4.log; tail -1 $O/c5.log
python3 tools/summarize_prof.py $O $O/all > /dev/null 2>&1 || true
ls $O
echo done
