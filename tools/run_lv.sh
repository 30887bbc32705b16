cd $GRAFT_REPO_ROOT
timeout 280 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for kind in text json; do
for lv in 1 2; do KIND=$kind LEVEL=$lv TAG=new timeout 120 python tools/level_time.py 2>&1 | tail -1; done
done
