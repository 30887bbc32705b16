#!/bin/bash
# Where does the compiler wait for memory?  Dumps, for one kernel of mlz_hip.hip, the sequence of global loads / stores,
# s_waitcnt vmcnt(N) and branches from the gfx950 assembly.  A load directly followed by "s_waitcnt vmcnt(0)" (typical
# for a load under a branch) is a serialized round trip; loads issued back to back with vmcnt(N > 0) waits are in flight
# together.  usage: tools/isa_waits.sh <kernel-name-substring> [extra -D flags]      (runs without a GPU)
k=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Xclang -target-feature -Xclang +unaligned-ds-access \
  -Wno-unused-command-line-argument "$@" -o /tmp/mlz_isa.s minlz_amd/csrc/mlz_hip.hip 2>&1 | grep -v "unaligned-ds"
for sym in $(grep -o "^_ZN3mlz[0-9]*[A-Za-z_0-9]*" /tmp/mlz_isa.s | sort -u | grep "$k"); do
  echo "== $sym"
  awk "/^${sym}/,/s_endpgm/" /tmp/mlz_isa.s | grep -n "global_load\|global_store\|s_waitcnt vmcnt\|s_cbranch\|s_barrier" | awk -F'\t' '{print $1 $2}'
done
