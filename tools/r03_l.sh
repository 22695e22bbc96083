set -e
cd $GRAFT_REPO_ROOT
python3 tools/ab_cycle.py --passes 4 --per-state 100 phases=1,local=1 phases=0,local=0 phases=1,local=0 phases=0,local=1
