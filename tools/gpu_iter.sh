cd $GRAFT_REPO_ROOT
timeout 900 python tools/qt_points_sweep.py 2>&1 | tail -12
