#!/bin/bash
# GPU box: encode-only kernel times of library variants (timing experiments whose output may be wrong). usage: tools/enc_variants.sh name...
cd $GRAFT_REPO_ROOT
for v in "$@" "$@"; do MINLZ_HIP_LIB=$GRAFT_REPO_ROOT/tools/var/$v.so python tools/enc_time.py 2>&1 | tail -1; done
