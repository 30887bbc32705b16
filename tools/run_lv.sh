#!/bin/bash
# Scratch driver used on the GPU box during the round: GPU test suite + encode/decode timing of the bench stream at
# levels 1 and 2 (text and JSON).  usage (from the repo root, via gpurun): bash tools/run_lv.sh
cd ${GRAFT_REPO_ROOT:-.}
timeout 280 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for k in text json; do for lv in 1 2; do KIND=$k LEVEL=$lv timeout 120 python tools/level_time.py 2>&1 | tail -1 | cut -c1-330; done; done
