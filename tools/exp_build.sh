#!/bin/bash
# Builds an experimental variant of the library: tools/exp_build.sh <name> [-DMACRO=value ...] -> tools/var/<name>.so
# (use with MINLZ_HIP_LIB=tools/var/<name>.so; tools/var/ is git-ignored but travels to the GPU box)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Xclang -target-feature -Xclang +unaligned-ds-access \
    -Wno-unused-command-line-argument "$@" -o tools/var/$name.so minlz_amd/csrc/mlz_hip.hip 2>&1 | grep -v "unaligned-ds-access" || true
ls -la tools/var/$name.so
