cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x
mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/md_trace --output-format csv -- python3 tools/md_bench.py --steps 100 > $O/md.log 2>&1
grep "ms/step" $O/md.log
