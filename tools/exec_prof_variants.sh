#!/bin/bash
# usage (GPU box): tools/exec_prof_variants.sh name1 name2 ...  — tools/gpu_prof.py (exec-pass phase cycles, wave 0 of each tile) with tools/var/<name>.so,
# built with tools/exp_build.sh <name> -DMLZ_PROFILE=1 [...]
cd $GRAFT_REPO_ROOT
for v in "$@"; do echo "== $v"; MINLZ_HIP_LIB=$GRAFT_REPO_ROOT/tools/var/$v.so python tools/gpu_prof.py 2>&1 | grep -v amdgpu.ids | tail -1; TAG=$v MINLZ_HIP_LIB=$GRAFT_REPO_ROOT/tools/var/$v.so python tools/dec_time.py 2>&1 | grep -v amdgpu.ids; done
