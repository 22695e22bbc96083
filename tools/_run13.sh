set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/md_stats --output-format csv -- python3 tools/md_bench.py --steps 300 > $O/md_bench.log 2> $O/md_bench.err || { tail -30 $O/md_bench.err; exit 1; }
tail -4 $O/md_bench.log
timeout -k 10 400 python3 tools/md_bench.py --steps 300 > $O/md_bench_noprof.log 2>&1
tail -4 $O/md_bench_noprof.log
echo done
