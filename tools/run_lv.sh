cd $GRAFT_REPO_ROOT
for kind in text json; do
KIND=$kind LEVEL=2 TAG=w1536 timeout 120 python tools/level_time.py 2>&1 | tail -1
done
timeout 280 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
