#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do
cp tools/libminlz_hip_$v.so minlz_amd/libminlz_hip.so
echo "== $v"; python tools/gpu_debug2.py 2>&1 | grep -v "ok$" | cut -c1-200 | head -4
done
