#!/bin/bash
# times the foreign-stream decode (8 MiB and 2 MiB blocks) with each experimental build given: tools/gen_sweep.sh g_a g_b ...
for v in "$@"; do
  for blk in 8388608 2097152; do
    echo -n "$v blk=$blk: "; BLOCK=$blk MINLZ_HIP_LIB=tools/var/$v.so timeout 300 python tools/foreign_time.py enwik 100 2>&1 | grep -v amdgpu.ids | tail -1 | sed -e 's/enwik 100 MB in//' -e "s/'dec_parse.*'dec_general'/'dec_general'/"
  done
done
