cd $GRAFT_REPO_ROOT
python3 tools/plan_cells_probe.py 2>&1 | tail -1 | cut -c1-60
python3 tools/plan_cells_probe.py --melt 100 2>&1 | tail -1 | cut -c1-60
python3 tools/plan_cells_probe.py 2>&1 | tail -1 | cut -c1-60
