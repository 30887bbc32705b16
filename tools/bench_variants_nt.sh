#!/bin/bash
cd $GRAFT_REPO_ROOT
cp minlz_amd/libminlz_hip.so /tmp/orig.so
for v in "$@"; do
  cp tools/libminlz_hip_$v.so minlz_amd/libminlz_hip.so
  python tools/gpu_prof.py 1 0 2>&1 | grep "cycles/step" | sed "s/^/$v /"
done
cp /tmp/orig.so minlz_amd/libminlz_hip.so
python tools/gpu_prof.py 1 0 2>&1 | grep "cycles/step" | sed "s/^/base /"
