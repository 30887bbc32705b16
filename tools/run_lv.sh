cd $GRAFT_REPO_ROOT
for k in text json; do KIND=$k LEVEL=1 TAG=tag timeout 120 python tools/level_time.py 2>&1 | tail -1 | cut -c1-200; done
timeout 280 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
