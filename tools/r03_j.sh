set -e
cd $GRAFT_REPO_ROOT
python3 tools/ab_cycle.py --passes 12 local=1 local=0
AZP_ROW_PHASES=0 python3 tools/ab_cycle.py --passes 6 local=1 local=0
