cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03t}
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -5 $O/tests.log
