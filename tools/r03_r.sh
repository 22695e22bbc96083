cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r
mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_domain.py -x -q > $O/dom.log 2>&1; echo "domain rc $?"; tail -25 $O/dom.log
