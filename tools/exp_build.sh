#!/bin/bash
# usage: exp_build.sh out.so [-DMACRO=1 ...]  — builds a kernel variant next to the product library
out=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Xclang -target-feature -Xclang +unaligned-ds-access -Wno-unused-command-line-argument "$@" -o $out minlz_amd/csrc/mlz_hip.hip 2>&1 | grep -v "unaligned-ds" | grep -E "error" 
