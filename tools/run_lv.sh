cd $GRAFT_REPO_ROOT
for k in text json; do for lv in 1 2; do KIND=$k LEVEL=$lv TAG=fartag timeout 120 python tools/level_time.py 2>&1 | tail -1 | cut -c1-210; done; done
timeout 280 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
