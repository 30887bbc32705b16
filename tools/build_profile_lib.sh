#!/bin/bash
# Builds build_var/libminlz_hip_prof.so with the in-kernel phase counters / tile timelines compiled in
# (-DMLZ_PROFILE=1).  tools/gpu_prof.py and tools/dec_trace.py need it:
#   bash tools/build_profile_lib.sh && MINLZ_HIP_LIB=$PWD/build_var/libminlz_hip_prof.so python tools/dec_trace.py
mkdir -p build_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Xclang -target-feature -Xclang +unaligned-ds-access \
  -Wno-unused-command-line-argument -DMLZ_PROFILE=1 "$@" -o build_var/libminlz_hip_prof.so minlz_amd/csrc/mlz_hip.hip 2>&1 | grep -v "unaligned-ds"
ls -la build_var/libminlz_hip_prof.so
