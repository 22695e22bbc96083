set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -2 $O/gpu_tests.log
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo done
