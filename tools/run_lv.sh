cd $GRAFT_REPO_ROOT
timeout 280 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for kind in text; do KIND=$kind LEVEL=1 TAG=$kind timeout 120 python tools/level_time.py 2>&1 | tail -1; done
