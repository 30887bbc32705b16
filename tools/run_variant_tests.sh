#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do
cp tools/libminlz_hip_$v.so minlz_amd/libminlz_hip.so
echo "== $v"; python -m pytest tests/test_gpu_encode.py tests/test_gpu_decode.py -x -q -m gpu 2>&1 | tail -3
done
